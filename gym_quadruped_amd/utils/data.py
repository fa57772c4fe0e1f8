"""Dataset export for batched rollouts - the reference's ``utils/data/h5py.py`` (H5Writer :90-172, H5Reader :175-230).

Layout (identical to the reference's file): group ``env_hparams`` holds the constructor arguments of the env
(``get_hyperparameters()``: lists / tuples JSON-encoded, class references as ``TYPE:module.Class``); group ``recordings``
holds one float64 dataset per observable plus ``action`` and ``time``, each of shape ``(trajectory, time, *obs_shape)``.
In a batch every env is one trajectory.

Recording is device-side: :class:`RolloutRecorder` keeps ``[T, N, dim]`` tensors on the env's GPU and copies nothing to the
host while the rollout runs (one ``tensor.copy_`` per observable and step, stream ordered); the host transfer happens once,
at export.  Export to HDF5 needs ``h5py`` (a third-party wheel, like in the reference, which imports it at module level -
absent in the build container, so that path is exercised only where it is installed); :meth:`RolloutRecorder.to_npz` writes
the same logical layout into a ``.npz`` archive (``recordings/<name>`` arrays + ``env_hparams`` as JSON) and
:func:`load_npz` reads it back.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch


def _hparams_to_jsonable(d):
    """The reference's save_dict_to_h5 conventions (h5py.py:22-48) applied to a plain dict."""
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out[k] = _hparams_to_jsonable(v)
        elif isinstance(v, (list, tuple)):
            if len(v) and isinstance(v[0], type):
                out[k] = [f'TYPE:{c.__module__}.{c.__name__}' for c in v]
            else:
                out[k] = [_hparams_to_jsonable(x) if isinstance(x, dict) else (x.tolist() if isinstance(x, np.ndarray) else x) for x in v]
        elif isinstance(v, np.ndarray):
            out[k] = v.tolist()
        elif isinstance(v, (str, int, float, bool)) or v is None:
            out[k] = v
        elif isinstance(v, (np.integer, np.floating)):
            out[k] = v.item()
        elif isinstance(v, torch.device):
            out[k] = str(v)
        else:
            raise TypeError(f"Cannot save type {type(v)} for key '{k}'")
    return out


class RolloutRecorder:
    """Records ``horizon`` steps of every env of a batched :class:`QuadrupedEnv` on the device.

        rec = RolloutRecorder(env, horizon=1000)
        obs = env.reset()
        for t in range(1000):
            action = policy(obs)
            obs, reward, terminated, truncated, info = env.step(action)
            rec.append(obs, action)
        rec.to_h5('rollouts.h5')          # or rec.to_npz('rollouts.npz')
    """

    def __init__(self, env, horizon: int, extra_obs: dict[str, tuple[int, ...]] | None = None):
        self.env = env
        self.horizon = int(horizon)
        self.t = 0
        _hparams_to_jsonable(env.get_hyperparameters())  # fail now, not at export time after the rollout was recorded
        n, dev = env.num_envs, env.device
        self.shapes = {k: tuple(sp.shape) for k, sp in env.observation_space.spaces.items()}
        self.shapes['action'] = tuple(env.action_space.shape)
        self.shapes.update({k: tuple(v) for k, v in (extra_obs or {}).items()})
        self.buf = {k: torch.empty((self.horizon, n) + shp, dtype=torch.float32, device=dev) for k, shp in self.shapes.items()}
        self.time = torch.empty((self.horizon, n), dtype=torch.float32, device=dev)

    def append(self, obs: dict, action, extra: dict | None = None):
        if self.t >= self.horizon:
            raise IndexError(f'recorder is full ({self.horizon} steps)')
        for k, v in obs.items():
            if k in self.buf:
                self.buf[k][self.t].copy_(torch.as_tensor(v, device=self.time.device).reshape(self.buf[k][self.t].shape))
        self.buf['action'][self.t].copy_(torch.as_tensor(action, device=self.time.device).reshape(self.buf['action'][self.t].shape))
        for k, v in (extra or {}).items():
            self.buf[k][self.t].copy_(torch.as_tensor(v, device=self.time.device).reshape(self.buf[k][self.t].shape))
        self.time[self.t].copy_(self.env.simulation_time if torch.is_tensor(self.env.simulation_time) else torch.as_tensor(self.env.simulation_time))
        self.t += 1

    def trajectories(self) -> dict[str, np.ndarray]:
        """``{name: float64 array (N, T, *shape)}`` of the recorded steps, plus ``time`` of shape ``(N, T, 1)``."""
        T = self.t
        out = {k: np.moveaxis(v[:T].double().cpu().numpy(), 0, 1) for k, v in self.buf.items()}
        out['time'] = np.moveaxis(self.time[:T].double().cpu().numpy(), 0, 1)[..., None]
        return out

    def to_npz(self, path):
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        arrays = {f'recordings/{k}': v for k, v in self.trajectories().items()}
        arrays['env_hparams'] = np.array(json.dumps(_hparams_to_jsonable(self.env.get_hyperparameters())))
        np.savez_compressed(path, **arrays)
        return path

    def to_h5(self, path):
        """The reference's HDF5 file (H5Writer + one append_trajectory per env, written in one go)."""
        try:
            import h5py
        except ImportError as e:   # same dependency as the reference (h5py.py:17)
            raise ImportError('to_h5 needs the h5py package; use to_npz for a dependency-free archive') from e
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        with h5py.File(path, 'w') as hf:
            g = hf.create_group('env_hparams')

            def save(group, d):
                for k, v in d.items():
                    if isinstance(v, dict):
                        save(group.require_group(k), v)
                    elif isinstance(v, list):
                        group.attrs[k] = json.dumps(v)
                    elif v is not None:
                        group.attrs[k] = v
            save(g, _hparams_to_jsonable(self.env.get_hyperparameters()))
            rec = hf.create_group('recordings')
            for k, v in self.trajectories().items():
                rec.create_dataset(k, data=v, maxshape=(None, None) + v.shape[2:], dtype='float64')
        return path


def load_npz(path):
    """``(recordings: {name: (N, T, *shape) float64}, env_hparams: dict)`` of a :meth:`RolloutRecorder.to_npz` archive."""
    z = np.load(path, allow_pickle=False)
    rec = {k.split('/', 1)[1]: z[k] for k in z.files if k.startswith('recordings/')}
    return rec, json.loads(str(z['env_hparams']))
