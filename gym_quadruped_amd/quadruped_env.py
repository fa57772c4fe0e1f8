"""Batched ``QuadrupedEnv`` - MI355X-native drop-in for the reference's ``gym_quadruped.quadruped_env.QuadrupedEnv``.

Same constructor arguments, method names, observable names (``ALL_OBS``) and return structure as the reference
(``gym_quadruped/quadruped_env.py:71-406``); every array gains a leading env axis ``N = num_envs`` and lives on the
GPU as a ``torch`` tensor.  Where the reference advances ONE MuJoCo env with ``mujoco.mj_step`` (:271) and assembles
observations in Python (:277-285), this class makes ONE call into ``libgq.so`` (``gq_step``, include/gq.h) that
advances all ``N`` envs, one env per wavefront, and writes observations / termination flags on the device.

PyTorch is plumbing here: it owns the device memory and the stream; all physics, observation, termination and
reset-randomisation arithmetic runs in the hand-written HIP kernels.  There is no CPU fallback.
"""
from __future__ import annotations

import copy
import ctypes as C
import logging
import math
from pathlib import Path
from typing import Any

import numpy as np
import torch

from . import _lib
from .accessors import AccessorsMixin
from .cabi import ALL_OBS as _ALL_OBS
from .cabi import LEG_NAMES, OBS_DIMS, GqObsOut, GqResampleCfg, GqResetCfg, GqState, MarshalledModel, obs_ids_from_names
from .mjcf import ModelDesc, compile_mjcf, load_compiled
from .robot_cfgs import RobotConfig, get_robot_config
from .terrain import generate_terrain
from .utils.math_utils import _process_range
from .utils.quadruped_utils import LegsAttr, configure_observation_space, extract_mj_joint_info, spaces

log = logging.getLogger(__name__)

BASE_OBS = _ALL_OBS[0:10]
BASE_OBS_BASE_FRAME = _ALL_OBS[10:15]
GEN_COORDS_OBS = _ALL_OBS[15:22]
FEET_OBS = _ALL_OBS[22:31]


def _make_info(env):
    """``info`` dict of ``step``: 'time', 'step_num' (value before this step's increment, as the reference reports
    it, quadruped_env.py:288-290) and 'invalid_contacts' (bool mask instead of a dict of MjContact objects).  A plain
    dict of persistent tensors: 'step_num' is refreshed in place by every step / reset, so ``get``, ``items``, ``**info``
    and copies all see it.  'contacts_dropped' has no reference counterpart: the number of contacts the narrow phase found in the
    step that did not enter the constraint set because the env was at the kernel's capacity (12 contacts / 63 rows; MuJoCo's
    mj_step has no such cap) - 0 on every env means the batch saw all its contacts."""
    return {'time': env._time, 'step_num': env._step_num_prev, 'invalid_contacts': env._invalid_b, 'contacts_dropped': env._contacts_dropped}


class QuadrupedEnv(AccessorsMixin):
    """Batched quadruped environment (see module docstring).  Single-env semantics follow the reference class of the
    same name; ``num_envs`` / ``device`` / ``auto_reset`` / solver knobs are the only additions."""

    _DEFAULT_OBS = ('qpos', 'qvel', 'tau_ctrl_setpoint', 'feet_pos:base', 'feet_vel:base')
    ALL_OBS = list(_ALL_OBS)
    metadata = {'render.modes': [], 'version': 0}

    def __init__(
        self,
        robot: str,
        state_obs_names: tuple[str, ...] = _DEFAULT_OBS,
        scene: str = 'flat',
        sim_dt: float = 0.002,
        base_vel_command_type: str = 'forward',
        ref_base_lin_vel: tuple[float, float] | float = 0.5,
        ref_base_ang_vel: tuple[float, float] | float = 0.0,
        ground_friction_coeff: tuple[float, float] | float = 1.0,
        legs_order: tuple[str, str, str, str] = ('FL', 'FR', 'RL', 'RR'),
        sensors: tuple = None,
        sensors_kwargs: tuple[dict[str, Any]] = None,
        external_disturbances_kwargs: dict[str, Any] = None,
        *,
        num_envs: int = 1,
        device: str | torch.device = 'cuda:0',
        auto_reset: bool | str = False,
        solver: str = 'newton',
        solver_iterations: int = 100,
        solver_tolerance: float = 1e-8,
        solver_noise_floor: float = 1e-5,
        seed: int | None = None,
        mjcf_path: str | None = None,
        env_id_offset: int = 0,
        accessors: bool = False,
        self_collision: bool | str | None = None,   # None / True / "convex": MuJoCo's behaviour, mesh pairs through the convex routine; "capsule": capsule proxies for mesh / cylinder pairs (faster, approximate); False: off
        pair_exchange: bool = True,   # convex self pairs of an entangled env are shared with idle wavefronts of the launch (csrc/gq_exchange.h): same results, shorter launches; False: every env keeps its pairs
    ):
        self._save_hyperparameters(constructor_params=locals().copy())
        log.info(f'Initializing {robot} environment with scene {scene}.')
        self.robot_name = robot
        self.robot_cfg: RobotConfig = get_robot_config(robot_name=robot)
        self.base_vel_command_type = base_vel_command_type
        self.base_lin_vel_range = _process_range(ref_base_lin_vel)
        self.base_ang_vel_range = _process_range(ref_base_ang_vel)
        self.ground_friction_coeff_range = _process_range(ground_friction_coeff)
        self.legs_order = tuple(legs_order)
        self.num_envs = int(num_envs)
        # False | True / 'same_step' (terminated envs are re-spawned inside the same step call; obs = first of the new
        # episode) | 'next_step' (gymnasium's NEXT_STEP: the terminal obs is returned and the env spends its next step
        # call on reset(), ignoring that action) - see gq_step in include/gq.h
        if auto_reset not in (False, True, 'same_step', 'next_step'):
            raise ValueError(f"auto_reset must be False, True, 'same_step' or 'next_step', got {auto_reset!r}")
        self.auto_reset = bool(auto_reset)
        self.auto_reset_mode = None if not auto_reset else ('next_step' if auto_reset == 'next_step' else 'same_step')
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.GqError('QuadrupedEnv runs on a ROCm GPU only (device must be cuda:N); there is no CPU path')

        # scene + model (reference :150-183)
        self.scene_desc, self.terrain_limits = generate_terrain(scene, self.robot_cfg.hip_height, seed=10)
        self.mjModel: ModelDesc = (compile_mjcf(mjcf_path) if mjcf_path
                                   else load_compiled(Path(self.robot_cfg.mjcf_filename).stem))
        qpos0 = self.mjModel.qpos0.copy()
        if self.robot_cfg.qpos0_js is not None:
            qpos0[7:] = np.asarray(self.robot_cfg.qpos0_js, dtype=np.float64)
        self.mjModel.qpos0 = qpos0
        self._mm = MarshalledModel(self.mjModel, qpos0=qpos0, feet_geom_names=self.robot_cfg.feet_geom_names,
                                   terrain_limits=self.terrain_limits, timestep=sim_dt, solver={'pgs': 0, 'newton': 1}[solver],
                                   iterations=solver_iterations, tolerance=solver_tolerance, noise_floor=solver_noise_floor,
                                   floor=self.scene_desc.get('floor'), boxes=self.scene_desc.get('boxes'),
                                   hfield=self.scene_desc.get('hfield'), self_collision=self_collision)
        self._sim_dt = float(sim_dt)

        # leg index maps (reference :189-212)
        self.joint_info = extract_mj_joint_info(self.mjModel)
        self.legs_qpos_idx = LegsAttr(None, None, None, None)
        self.legs_qvel_idx = LegsAttr(None, None, None, None)
        self.legs_tau_idx = LegsAttr(None, None, None, None)
        for leg in ['FR', 'FL', 'RR', 'RL']:
            qi, vi, ti = [], [], []
            for jn in self.robot_cfg.leg_joints[leg]:
                assert jn in self.joint_info, f'Joint {jn} not found in {list(self.joint_info.keys())}'
                qi.extend(self.joint_info[jn].qpos_idx); vi.extend(self.joint_info[jn].qvel_idx); ti.extend(self.joint_info[jn].tau_idx)
            self.legs_qpos_idx[leg], self.legs_qvel_idx[leg], self.legs_tau_idx[leg] = qi, vi, ti
        self._feet_geom_id = LegsAttr(None, None, None, None)
        self._feet_body_id = LegsAttr(None, None, None, None)
        for leg in ['FR', 'FL', 'RR', 'RL']:
            g = self.mjModel.geom_names.index(self.robot_cfg.feet_geom_names[leg])
            self._feet_geom_id[leg] = g
            self._feet_body_id[leg] = int(self.mjModel.geom_bodyid[g])

        # spaces (reference :215-230; the action Box is unbounded there because of the truthiness slip, quirk B2)
        nu = self.mjModel.nu
        self.action_space = spaces.Box(shape=(nu,), low=np.full(nu, -np.inf), high=np.full(nu, np.inf), dtype=np.float32)
        self.state_obs_names = tuple(state_obs_names)
        self.observation_space = configure_observation_space(mj_model=self.mjModel, obs_names=self.state_obs_names)
        self._obs_ids = obs_ids_from_names(self.state_obs_names)
        # accessors=True: the kernel also writes every ALL_OBS observable the user did not ask for, behind the user's
        # columns, so that the reference's getters (base_lin_vel(frame), feet_pos(frame), ...) are views (accessors.py)
        self._extra_names = tuple(n for n in _ALL_OBS if n not in self.state_obs_names) if accessors else ()
        self._all_ids = list(self._obs_ids) + obs_ids_from_names(self._extra_names)
        self._launches = 0
        self._hm_fresh = False   # a HeightMap(follow_base=True) holds the rays of the CURRENT state (written by the last step's kernel)
        self._hm_follow = None   # weakref to THE HeightMap(follow_base=True) registered with the step kernel (one output slot per batch)
        self._hm_version = -1

        # device state: one tensor per field, env-major rows
        N, dev = self.num_envs, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self._qpos = torch.zeros(N, 19, dtype=torch.float64, device=dev)
        self._qvel = torch.zeros(N, 18, **f32)
        self._qacc = torch.zeros(N, 18, **f32)
        self._warm = torch.zeros(N, 18, **f32)
        self._applied = torch.zeros(N, 18, **f32)
        self._time = torch.zeros(N, **f32)
        self._friction = torch.full((N,), -1.0, **f32)   # < 0: XML frictions until the first reset sets it
        self._cmd = torch.zeros(N, 4, **f32)
        self._ctrl = torch.zeros(N, nu, **f32)
        self._last_action = self._ctrl
        self._obs_dim = int(sum(OBS_DIMS[i] for i in self._all_ids))   # row width the kernel writes
        self._obs_buf = torch.zeros(N, self._obs_dim, **f32)
        self._reward = torch.zeros(N, **f32)
        self._terminated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._truncated = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._invalid = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._terminated_b = self._terminated.view(torch.bool)
        self._truncated_b = self._truncated.view(torch.bool)
        self._invalid_b = self._invalid.view(torch.bool)
        self._step_num = torch.zeros(N, dtype=torch.int32, device=dev)
        self._step_num_prev = torch.zeros(N, dtype=torch.int32, device=dev)   # info['step_num'], written by the kernel
        self._contacts_dropped = torch.zeros(N, dtype=torch.int32, device=dev)  # info['contacts_dropped'], written by the kernel
        self._lift_failed = torch.zeros(N, dtype=torch.uint8, device=dev)
        self._mask_all = torch.ones(N, dtype=torch.uint8, device=dev)
        # in-episode resampling state (reference :292-305), advanced by the step kernel's epilogue:
        # columns {after_vel, before_vel, n_vel, after_dist, before_dist, n_dist} (gq_batch_set_resampling)
        self._h9 = torch.zeros(N, 6, dtype=torch.int32, device=dev)
        self._h9[:, 1] = 1 << 30
        self._h9[:, 4] = 1 << 30
        self._ext_dist = torch.zeros(N, 6, **f32)
        self._has_cmd = False
        self._obs_views, self._extra_views, k = {}, {}, 0
        for name, i in zip(self.state_obs_names + self._extra_names, self._all_ids):
            (self._obs_views if name in self.state_obs_names else self._extra_views)[name] = self._obs_buf[:, k:k + OBS_DIMS[i]]
            k += OBS_DIMS[i]
        self._key_qpos = torch.as_tensor(self.mjModel.key_qpos[0] if len(self.mjModel.key_qpos) else qpos0, dtype=torch.float64, device=dev)
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(0 if seed is None else int(seed))

        # C-ABI handles
        L = _lib.lib()
        self._L = L
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._hmodel = C.c_void_p()
        _lib.check(L.gq_model_create(C.byref(self._mm.desc), int(dev_index), C.byref(self._hmodel)), 'gq_model_create')
        ids = np.asarray(self._all_ids, dtype=np.int32)
        lo = np.asarray([LEG_NAMES.index(l) for l in self.legs_order], dtype=np.int32)
        self._hbatch = C.c_void_p()
        _lib.check(L.gq_batch_create(self._hmodel, N, ids.ctypes.data, len(ids), lo.ctypes.data, C.byref(self._hbatch)), 'gq_batch_create')
        assert L.gq_batch_obs_dim(self._hbatch) == self._obs_dim
        if not pair_exchange and self._mm.self_collision == 'convex':
            L.gq_batch_set_pair_exchange(self._hbatch, 0)   # (a model without convex self pairs has no exchange to switch off)
        self._st = GqState(self._qpos.data_ptr(), self._qvel.data_ptr(), self._qacc.data_ptr(), self._warm.data_ptr(),
                           self._applied.data_ptr(), self._time.data_ptr(), self._friction.data_ptr(), self._cmd.data_ptr())
        self._out = GqObsOut(self._obs_buf.data_ptr(), self._reward.data_ptr(), self._terminated.data_ptr(),
                             self._truncated.data_ptr(), self._invalid.data_ptr(), self._step_num.data_ptr(),
                             self._step_num_prev.data_ptr(), self._contacts_dropped.data_ptr())
        # accessors=True: the production kernel also writes the dynamics row (mj_fullM, qfrc_bias, body poses, foot points) and
        # the contact row (mjData.contact + mj_contactForce) of every step - what the reference's model-based-control getters
        # read from mjData (accessors.py); no instrumented kernel variant involved
        self._dyn = self._contacts = None
        if accessors:
            from .cabi import GQ_CON_STRIDE, GQ_DYN
            self._dyn = torch.zeros(N, GQ_DYN['STRIDE'], **f32)
            self._contacts = torch.zeros(N, GQ_CON_STRIDE, **f32)
            _lib.check(L.gq_batch_set_outputs(self._hbatch, self._dyn.data_ptr(), self._contacts.data_ptr()), 'gq_batch_set_outputs')
        self._info = _make_info(self)
        self._episode = torch.zeros(N, dtype=torch.int32, device=dev)
        t = self.base_vel_command_type
        if not any(k in t for k in ('forward', 'random', 'human')):
            raise ValueError(f'Invalid base linear velocity command type: {t}')
        self._seed = 0 if seed is None else int(seed)
        self._reset_cfg = GqResetCfg(
            seed=self._seed, random=1, q_pos_amp=20 * math.pi / 180, q_vel_amp=0.5, roll_sweep=10 * math.pi / 180,
            pitch_sweep=10 * math.pi / 180, hip_height=float(self.robot_cfg.hip_height),
            lin_vel_range=(C.c_float * 2)(*map(float, self.base_lin_vel_range)),
            ang_vel_range=(C.c_float * 2)(*map(float, self.base_ang_vel_range)),
            friction_range=(C.c_float * 2)(*map(float, self.ground_friction_coeff_range)),
            cmd_forward=int('forward' in t), cmd_random=int('forward' not in t and 'random' in t),
            cmd_rotate=int('rotate' in t), cmd_human=int('human' in t), env_id_offset=int(env_id_offset))
        # auto-reset happens inside the step kernel (terminated envs take a second pass); None = off
        self._auto_cfg_struct = GqResetCfg.from_buffer_copy(self._reset_cfg)  # user reset() options never leak into it
        self._auto_cfg_struct.autoreset_next_step = int(self.auto_reset_mode == 'next_step')
        self._auto_cfg = C.pointer(self._auto_cfg_struct) if self.auto_reset else None

        self.external_disturbances_kwargs = external_disturbances_kwargs
        if self.external_disturbances_kwargs is not None:
            self._sample_external_disturbances(self._mask_all.view(torch.bool))   # reference :240-242 (constructor draw)
        dist_reset = external_disturbances_kwargs is not None and external_disturbances_kwargs.get('type') == 'reset'
        if 'reset' in t or dist_reset:
            rs = GqResampleCfg(seed=self._seed, cmd_reset=int('reset' in t), dist_reset=int(dist_reset), env_id_offset=int(env_id_offset))
            for k, key in enumerate(('x', 'y', 'z', 'roll', 'pitch', 'yaw')):
                r = (external_disturbances_kwargs or {}).get(key)
                kind = 0 if (r is None or len(r) == 0) else min(len(r), 2)
                rs.dist_kind[k] = kind
                rs.dist_range[k][0] = float(r[0]) if kind >= 1 else 0.0
                rs.dist_range[k][1] = float(r[1]) if kind == 2 else 0.0
            self._resample_cfg = rs
            _lib.check(L.gq_batch_set_resampling(self._hbatch, C.byref(rs), C.byref(self._reset_cfg), self._h9.data_ptr(),
                                                 self._ext_dist.data_ptr()), 'gq_batch_set_resampling')
        self.viewer = None
        # sensors (reference :232-236): sensor_cls(mj_model=..., mj_data=..., **kwargs); mj_data is this env
        self.sensors = []
        if sensors is not None:
            for sensor_cls, kw in zip(sensors, sensors_kwargs or [{}] * len(sensors)):
                self.sensors.append(sensor_cls(mj_model=self.mjModel, mj_data=self, **kw))
        from .sensors.imu import IMU
        for sn in self.sensors:
            if isinstance(sn, IMU):
                sn.cfg.seed = int(sn.cfg.seed) + int(env_id_offset)
                _lib.check(L.gq_batch_set_imu(self._hbatch, C.byref(sn.cfg), sn.bias_state.data_ptr()), 'gq_batch_set_imu')
        sensor_obs = [o for sn in self.sensors for o in sn.available_observations()]
        for name in self.state_obs_names:
            if name.startswith('imu') and name not in sensor_obs:
                raise ValueError(f'Invalid observation name: {name}: no sensor provides it (pass sensors=(IMU,), sensors_kwargs=...)')
        self._profile_events = None  # optional (start, end) torch.cuda.Event pair recorded around the gq_step launch

    # ------------------------------------------------------------------ core API
    def step(self, action):
        """Advance every env by one ``sim_dt`` (reference ``step`` :251-307).

        action: ``[N, nu]`` joint torques (tensor / array; ``[nu]`` is broadcast when N == 1).
        Returns ``(obs, reward, terminated, truncated, info)`` with obs a dict ``name -> [N, dim]`` float32 tensor
        (views of one persistent buffer, overwritten by the next call), reward ``[N]``, terminated/truncated ``[N]``
        bool and info {'time' [N], 'step_num' [N], 'invalid_contacts' [N] bool}.
        """
        if (torch.is_tensor(action) and action.dtype == torch.float32 and action.device == self.device
                and action.shape == self._ctrl.shape and action.is_contiguous()):
            self._last_action = action      # zero-copy: the kernel reads the caller's tensor
        else:
            a = torch.as_tensor(action, dtype=torch.float32, device=self.device)
            if a.dim() == 1:
                a = a.unsqueeze(0).expand(self.num_envs, -1)
            if a.shape != self._ctrl.shape:
                raise ValueError(f'action must have shape {tuple(self._ctrl.shape)}, got {tuple(a.shape)}')
            self._ctrl.copy_(a)
            self._last_action = self._ctrl
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ev = self._profile_events
        if ev is not None:
            ev[0].record()
        _lib.check(self._L.gq_step(self._hbatch, self._last_action.data_ptr(), None, self._st, self._out, self._auto_cfg,
                                   self._episode.data_ptr(), self._lift_failed.data_ptr(), stream), 'gq_step')
        if ev is not None:
            ev[1].record()
        self._launches += 1
        # a HeightMap(follow_base=True) registered with this env now holds the rays of the NEW state; the flag is tied to the version counter of the
        # state tensor, so that an in-place write (env.qpos[...] = x) makes it stale again
        self._hm_fresh = self._hm_follow is not None and self._hm_follow() is not None
        self._hm_version = self._qpos._version
        self._note_step()
        for sensor in self.sensors:  # reference :273-274 (kernel-side sensors: no-op)
            sensor.step()

        # command / disturbance resampling (reference :292-305) ran in the kernel's epilogue (gq_batch_set_resampling)
        return self._obs_views, self._reward, self._terminated_b, self._truncated_b, self._info

    def rollout(self, actions, shards: int = 2, obs_out=None):
        """Open-loop rollout: ``K = len(actions)`` steps of every env with the given action sequence (``[K, N, nu]`` float32 on
        the device), equivalent to ``for a in actions: env.step(a)`` - same kernels, same state afterwards, bit for bit -
        but pipelined: the batch is cut into ``shards`` contiguous groups of envs, each group's K launches are chained on a HIP
        stream of its own, and since envs are independent no launch waits for another group's stragglers or for the gap
        between two dependent launches (``gq_step_range``).  With a policy in the loop this is not available - the next
        action needs every env's observation - which is what ``step`` is for; random-action rollouts, dataset recording
        (the reference's examples/aliengo_dataset.py) and replaying planned torque sequences are.  Measured gain on MI355X at
        4096 envs: +6 % with 2 shards; more shards lose (overlapping launches share the SIMDs).

        ``obs_out``: optional ``[K, N, obs_dim]`` float32 tensor that receives the observation rows of every step.
        Host-side sensors get their K ``step()`` calls after the launches (they see the final state only); the per-step
        profiling events of ``step`` are not recorded.
        Returns the observation dict of the last step (views, like ``step``)."""
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device)
        if a.dim() != 3 or tuple(a.shape[1:]) != tuple(self._ctrl.shape):
            raise ValueError(f'actions must have shape (K, {self.num_envs}, {self.mjModel.nu}), got {tuple(a.shape)}')
        a = a.contiguous()
        K = int(a.shape[0])
        if K == 0:
            raise ValueError('rollout needs at least one step (actions has K == 0)')
        if obs_out is not None and (tuple(obs_out.shape) != (K, self.num_envs, self._obs_dim) or obs_out.dtype != torch.float32 or not obs_out.is_contiguous()):
            raise ValueError(f'obs_out must be a contiguous float32 tensor of shape {(K, self.num_envs, self._obs_dim)}')
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._hm_fresh = False
        _lib.check(self._L.gq_rollout(self._hbatch, a.data_ptr(), K, int(shards), self._st, self._out, self._auto_cfg, self._episode.data_ptr(),
                                      self._lift_failed.data_ptr(), None if obs_out is None else obs_out.data_ptr(), stream), 'gq_rollout')
        self._last_action = a[K - 1]
        self._launches += K
        self._note_step()
        for sensor in self.sensors:  # host-side sensors advance once per env step, as in the step loop (kernel-side ones: no-op)
            for _ in range(K):
                sensor.step()
        return self._obs_views

    def rollout_closed_loop(self, n_steps: int, kp, kd, q_des=None, *, mode: str = 'inline', record_obs: bool = False, record_actions: bool = False,
                            noise_sigma: float = 0.0, policy_waves: int = 0, step_waves: int = 0, timeout_s: float = 5.0, check: bool = True):
        """``n_steps`` steps of every env with a joint-space PD policy IN the loop and no launch boundary (``gq_rollout_closed``):
        the device-side form of ``for k: a = kp * (q_des - obs['qpos_js']) - kd * obs['qvel_js']; obs, ... = env.step(a)``
        (the reference's control loop, README.md:31-33, around quadruped_env.py:251-307).  Env-steps are tasks: a policy kernel on
        a second stream turns each published observation row into the env's next action and queues the env; the wavefronts of
        one persistent step launch pop ready envs, step them (next-step auto-reset included) and publish the result.  An env
        waits for ITS action only - never for the slowest env of a step - so the throughput is that of the open-loop
        persistent rollout, with the loop closed.  State, flags and observations afterwards equal the step loop's with the same
        actions, bit for bit.

        mode 'mailbox': the policy is a kernel of its own on a second stream, actions and observations travel through per-env
        mailboxes and per-XCD ready queues - the general mechanism (any resident policy kernel can take that seat); 'inline': the
        wavefront that steps an env evaluates the PD law itself - no turn-around latency, the faster form when there are no more
        envs than wavefront slots (4096 on an MI355X).  Both leave the same bits.

        kp, kd: scalars or 12 values (hinge order of ``qpos[7:]``); q_des: 12 joint angles (default: keyframe 0); noise_sigma:
        Gaussian exploration noise added to every torque (counter-based draws keyed by the env's seed, the step and the joint).
        Returns a dict with the last observation views under 'obs' and, when asked, 'obs_seq' ``[K, N, obs_dim]`` /
        'actions' ``[K, N, 12]``.  ``check``: wait for the rollout and raise ``GqError`` if a participant gave up waiting
        (deadline ``timeout_s`` per wait) instead of leaving that to the caller."""
        from .cabi import GqPolicyPd
        K = int(n_steps)
        if K < 0:
            raise ValueError('n_steps must be >= 0')
        qd = self._key_qpos[7:19].float().cpu().numpy() if q_des is None else np.asarray(q_des, dtype=np.float32).reshape(12)
        pd = GqPolicyPd()
        pd.kp = (C.c_float * 12)(*np.broadcast_to(np.asarray(kp, dtype=np.float32), (12,)))
        pd.kd = (C.c_float * 12)(*np.broadcast_to(np.asarray(kd, dtype=np.float32), (12,)))
        pd.q_des = (C.c_float * 12)(*[float(v) for v in qd])
        pd.noise_sigma, pd.noise_seed, pd.noise_step0 = float(noise_sigma), int(self._seed or 0) & (2 ** 64 - 1), int(self._launches) & 0x7fffffff
        f32 = dict(dtype=torch.float32, device=self.device)
        obs_seq = torch.empty(K, self.num_envs, self._obs_dim, **f32) if record_obs else None
        act_seq = torch.empty(K, self.num_envs, self.mjModel.nu, **f32) if record_actions else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if mode not in ('inline', 'mailbox'):
            raise ValueError("mode must be 'inline' or 'mailbox'")
        self._hm_fresh = False
        _lib.check(self._L.gq_rollout_closed(self._hbatch, K, int(mode == 'inline'), C.byref(pd), int(policy_waves), int(step_waves), float(timeout_s), self._st, self._out,
                                             self._auto_cfg, self._episode.data_ptr(), self._lift_failed.data_ptr(),
                                             None if obs_seq is None else obs_seq.data_ptr(), None if act_seq is None else act_seq.data_ptr(), stream),
                   'gq_rollout_closed')
        self._launches += K
        if check:
            self.closed_loop_status()
        for sensor in self.sensors:
            for _ in range(K):
                sensor.step()
        return {'obs': self._obs_views, 'obs_seq': obs_seq, 'actions': act_seq}

    def closed_loop_status(self):
        """Wait for the last ``rollout_closed_loop`` and return (abort code, detail, env-steps played); raises ``GqError`` if it
        was aborted (a wait passed its deadline: missing / stuck policy, or the policy kernel could not become resident)."""
        st = (C.c_int32 * 4)()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.gq_rollout_closed_status(self._hbatch, st, stream), 'gq_rollout_closed')
        return int(st[0]), int(st[1]), int(st[2])

    def reset(self, qpos=None, qvel=None, seed: int | None = None, random: bool = True,
              options: dict[str, Any] | None = None, env_ids=None):
        """Reset (reference ``reset`` :309-406); ``env_ids`` (index tensor / bool mask) restricts it to a subset.
        Returns the observation dict only, like the reference (quirk B7)."""
        if seed is not None:  # reference: np.random.seed(seed) (:338-339); here: re-key the device generators
            self._gen.manual_seed(int(seed))
            self._seed = int(seed)
            self._reset_cfg.seed = self._seed
            self._auto_cfg_struct.seed = self._seed
            self._episode.zero_()
            if getattr(self, '_resample_cfg', None) is not None:
                self._resample_cfg.seed = self._seed
                self._h9[:, 2] = 0; self._h9[:, 5] = 0
                _lib.check(self._L.gq_batch_set_resampling(self._hbatch, C.byref(self._resample_cfg), C.byref(self._reset_cfg),
                                                           self._h9.data_ptr(), self._ext_dist.data_ptr()), 'gq_batch_set_resampling')
        N = self.num_envs
        if env_ids is None:
            mask = self._mask_all
        else:
            idx = torch.as_tensor(env_ids, device=self.device)
            if idx.dtype == torch.bool:
                mask = idx.to(torch.uint8)
            else:
                mask = torch.zeros(N, dtype=torch.uint8, device=self.device)
                mask[idx.long()] = 1
        if qpos is None and qvel is None:
            self._reset_masked(mask, random=random, options=options)
        else:
            qp = torch.as_tensor(qpos, dtype=torch.float64, device=self.device).reshape(-1, 19).expand(N, 19).contiguous()
            qv = torch.as_tensor(qvel, dtype=torch.float32, device=self.device).reshape(-1, 18).expand(N, 18).contiguous()
            self._reset_masked(mask, random=False, options=options, qpos=qp, qvel=qv)
        return self._obs_views

    def _reset_masked(self, mask, random, options, qpos=None, qvel=None):
        """One ``gq_reset`` call = state write (+ lift loop) and the reset's own ``mj_step`` for the masked envs."""
        options = {} if options is None else options
        self._hm_fresh = False
        cfg = self._reset_cfg
        cfg.random = int(bool(random))
        cfg.q_pos_amp = float(options.get('angle_sweep', 20 * math.pi / 180))
        cfg.roll_sweep = float(options.get('roll_sweep', 10 * math.pi / 180))
        cfg.pitch_sweep = float(options.get('pitch_sweep', 10 * math.pi / 180))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.gq_reset(self._hbatch, mask.data_ptr(), None if qpos is None else qpos.data_ptr(),
                                    None if qvel is None else qvel.data_ptr(), C.byref(cfg), self._st, self._out,
                                    self._episode.data_ptr(), self._lift_failed.data_ptr(), stream), 'gq_reset')
        self._launches += 1
        self._note_step()
        if mask is self._mask_all:  # the reference zeroes mjData.ctrl in reset (:334): torque_ctrl_setpoint reads zero afterwards
            self._ctrl.zero_()
            self._last_action = self._ctrl

    # ------------------------------------------------------------------ command / disturbance sampling
    def _sample_ref_vel(self, mask):
        """Per-env redraw of the velocity command for envs in ``mask`` (reference ``_sample_ref_vel`` :1046-1072)."""
        N, dev, g = self.num_envs, self.device, self._gen
        t = self.base_vel_command_type
        lo, hi = self.base_lin_vel_range
        if 'forward' in t:
            norm = lo + (hi - lo) * torch.rand(N, generator=g, device=dev)
            heading = torch.zeros(N, device=dev)
        elif 'random' in t:
            norm = lo + (hi - lo) * torch.rand(N, generator=g, device=dev)
            heading = (torch.rand(N, generator=g, device=dev) * 2 - 1) * math.pi
        elif 'human' in t:
            norm = torch.zeros(N, device=dev)
            heading = torch.zeros(N, device=dev)
        else:
            raise ValueError(f'Invalid base linear velocity command type: {t}')
        if 'rotate' in t:
            alo, ahi = self.base_ang_vel_range
            yaw_dot = alo + (ahi - alo) * torch.rand(N, generator=g, device=dev)
        else:
            yaw_dot = torch.zeros(N, device=dev)
        new = torch.stack([norm * torch.cos(heading), norm * torch.sin(heading), torch.zeros(N, device=dev), yaw_dot], 1)
        self._cmd.copy_(torch.where(mask.unsqueeze(1), new, self._cmd))
        if 'reset' in t:
            nxt = torch.randint(1000, 3000, (N,), generator=g, device=dev, dtype=torch.int32)
            self._h9[:, 1] = torch.where(mask, nxt, self._h9[:, 1])
            self._h9[:, 0] = torch.where(mask, torch.zeros_like(nxt), self._h9[:, 0])
        self._has_cmd = True

    def _sample_external_disturbances(self, mask):
        """Per-env redraw of the base wrench (reference ``_sample_external_disturbances`` :1074-1139)."""
        N, dev, g = self.num_envs, self.device, self._gen
        kw = self.external_disturbances_kwargs
        cols = []
        for key in ('x', 'y', 'z', 'roll', 'pitch', 'yaw'):
            r = kw.get(key)
            if r is None or len(r) == 0:
                cols.append(torch.zeros(N, device=dev))
            elif len(r) == 1:
                cols.append(torch.full((N,), float(r[0]), device=dev))
            else:
                cols.append(float(r[0]) + (float(r[1]) - float(r[0])) * torch.rand(N, generator=g, device=dev))
        new = torch.stack(cols, 1)
        self._ext_dist.copy_(torch.where(mask.unsqueeze(1), new, self._ext_dist))   # in place: the kernel holds the pointer
        nxt = torch.randint(1000, 3000, (N,), generator=g, device=dev, dtype=torch.int32)
        self._h9[:, 4] = torch.where(mask, nxt, self._h9[:, 4])
        self._h9[:, 3] = torch.where(mask, torch.zeros_like(nxt), self._h9[:, 3])

    # ------------------------------------------------------------------ accessors (reference names)
    @property
    def qpos(self):
        return self._qpos

    @property
    def qvel(self):
        return self._qvel

    @property
    def base_pos(self):
        return self._qpos[:, 0:3]

    @property
    def joint_space_state(self):
        return self._qpos[:, 7:], self._qvel[:, 6:]

    @property
    def torque_ctrl_setpoint(self):
        return self._last_action

    @property
    def simulation_dt(self):
        return self._sim_dt

    @property
    def simulation_time(self):
        return self._time

    @property
    def step_num(self):
        return self._step_num

    @property
    def robot_model(self):
        return self.mjModel

    @property
    def lift_failed(self):
        """Per-env flag of the reset RuntimeError condition (reference :387-388)."""
        return self._lift_failed.view(torch.bool)

    def target_base_vel(self):
        """Reference command in the heading frame ``[N,3]`` and yaw rate ``[N]`` (reference :488-499 inputs)."""
        return self._cmd[:, 0:3], self._cmd[:, 3]

    def state_dict(self):
        """Checkpoint: everything needed to resume a rollout bit-for-bit."""
        keys = ['_qpos', '_qvel', '_qacc', '_warm', '_applied', '_time', '_friction', '_cmd', '_step_num', '_episode', '_terminated',
                '_h9', '_ext_dist', '_step_num_prev']
        keys.append('_lift_failed')
        d = {k: getattr(self, k).clone() for k in keys}
        d['rng'] = self._gen.get_state()
        # sensor state the kernel reads and writes every step: the IMU bias random walks
        d['sensor_bias'] = [sn.bias_state.clone() if hasattr(sn, 'bias_state') else None for sn in self.sensors]
        return d

    _LEGACY_H9 = {'_steps_after_vel': 0, '_steps_before_vel': 1, '_steps_after_dist': 3, '_steps_before_dist': 4}

    def load_state_dict(self, d):
        self._hm_fresh = False
        for k, v in d.items():
            if k in self._LEGACY_H9:  # checkpoints written before the resampling counters moved into one [N, 6] tensor
                self._h9[:, self._LEGACY_H9[k]].copy_(torch.as_tensor(v, device=self.device).to(torch.int32))
                continue
            if k == 'rng':
                self._gen.set_state(v)
            elif k == 'sensor_bias':
                for sn, b in zip(self.sensors, v):
                    if b is not None:
                        sn.bias_state.copy_(b)  # in place: the kernel holds this tensor's pointer
            else:
                getattr(self, k).copy_(v)
        if '_terminated' in d:  # next-step auto-reset: the envs that terminated last step are still waiting for their reset
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._L.gq_batch_set_pending(self._hbatch, self._terminated.data_ptr(), stream), 'gq_batch_set_pending')

    def debug_internals(self, n_envs: int, names):
        """Copy solver / dynamics internals of the LAST step for the first ``n_envs`` envs (must be enabled before
        the step with ``enable_debug``)."""
        out = []
        for e in range(n_envs):
            rec = {}
            for nm in names:
                buf = np.zeros(64 * 18, dtype=np.float64)
                n = _lib.check(self._L.gq_debug_get(self._hbatch, e, nm.encode(), buf.ctypes.data, buf.size), 'gq_debug_get')
                rec[nm] = buf[:n].copy()
            out.append(rec)
        return out

    def enable_debug(self, n_envs: int):
        _lib.check(self._L.gq_debug_enable(self._hbatch, int(n_envs)), 'gq_debug_enable')

    def render(self, *a, **k):
        raise NotImplementedError('interactive MuJoCo viewer rendering is out of scope of the batched GPU path')

    def close(self):
        if getattr(self, '_hbatch', None):
            self._L.gq_batch_destroy(self._hbatch)
            self._hbatch = None
        if getattr(self, '_hmodel', None):
            self._L.gq_model_destroy(self._hmodel)
            self._hmodel = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _save_hyperparameters(self, constructor_params):
        self._init_args = constructor_params
        for k in ['self', '__class__']:
            self._init_args.pop(k, None)

    def get_hyperparameters(self):
        """Constructor arguments (reference :1356-1358)."""
        return copy.copy(self._init_args)

    def __str__(self):
        msg = f'robot={self._init_args["robot"]} terrain={self._init_args["scene"]} task={self.base_vel_command_type} num_envs={self.num_envs}'
        if self.base_vel_command_type != 'human':
            msg += (f' lin_vel_range=({self.base_lin_vel_range[0]:.3f}, {self.base_lin_vel_range[1]:.3f})'
                    f' ang_vel_range=({self.base_ang_vel_range[0]:.3f}, {self.base_ang_vel_range[1]:.3f})'
                    f' lat_friction_range=({self.ground_friction_coeff_range[0]:.1e}, {self.ground_friction_coeff_range[1]:.1e})')
        return msg
