"""ctypes wrapper of oracle/libgq_oracle.so (CPU fp64 single-env restatement of the reference step).

TEST INFRASTRUCTURE ONLY - see oracle/gq_oracle.c header ("parity unpinned" for the mj_step part).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from gym_quadruped_amd.cabi import GqModelDesc, MarshalledModel, obs_ids_from_names, OBS_DIMS

_HERE = Path(__file__).parent
_LIB = None


def build(force=False):
    so = _HERE / 'libgq_oracle.so'
    srcs = [_HERE / 'gq_oracle.c', _HERE / 'gq_convex.h', _HERE.parent / 'include' / 'gq.h']
    if force or not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in srcs):
        subprocess.run(['make', '-C', str(_HERE), '-B' if force else '-s'], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / 'libgq_oracle.so'
        if not so.exists():
            build()
        L = C.CDLL(str(so))
        L.gqo_create.argtypes = [C.POINTER(GqModelDesc), C.POINTER(C.c_void_p)]
        L.gqo_destroy.argtypes = [C.c_void_p]
        L.gqo_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.gqo_step.argtypes = [C.c_void_p, C.c_void_p]
        L.gqo_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_double, C.c_double]
        L.gqo_set_solver.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.gqo_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.gqo_jac_point.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.gqo_get_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gqo_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.gqo_set_imu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.gqo_primal_cost.argtypes = [C.c_void_p, C.c_void_p]
        L.gqo_primal_cost.restype = C.c_double
        L.gqo_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One CPU env.  Field names follow mujoco.MjData."""

    def __init__(self, mm: MarshalledModel):
        self.mm = mm
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.gqo_create(C.byref(mm.desc), C.byref(h))
        if rc != 0:
            raise RuntimeError(self.L.gqo_last_error().decode())
        self.h = h
        self.nv, self.nq, self.nu = mm.md.nv, mm.md.nq, mm.md.nu

    def __del__(self):
        if getattr(self, 'h', None):
            self.L.gqo_destroy(self.h)
            self.h = None

    def set_state(self, qpos=None, qvel=None, qacc_warmstart=None, qfrc_applied=None, time=0.0, friction=-1.0):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (qpos, qvel, qacc_warmstart, qfrc_applied)]
        self.L.gqo_set_state(self.h, *[_p(a) for a in arrs], float(time), float(friction))

    def set_imu(self, pos=(0, 0, 0), quat=(1, 0, 0, 0)):
        p, q = np.asarray(pos, dtype=np.float64), np.asarray(quat, dtype=np.float64)
        self.L.gqo_set_imu(self.h, _p(p), _p(q))

    def set_solver(self, solver, iterations=100, tolerance=1e-8):
        self.L.gqo_set_solver(self.h, int(solver), int(iterations), float(tolerance))

    def primal_cost(self, qacc):
        a = np.ascontiguousarray(qacc, dtype=np.float64)
        return float(self.L.gqo_primal_cost(self.h, _p(a)))

    def forward(self, ctrl=None, stage=0):
        c = None if ctrl is None else np.ascontiguousarray(ctrl, dtype=np.float64)
        self.L.gqo_forward(self.h, _p(c), stage)

    def step(self, ctrl):
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        self.L.gqo_step(self.h, _p(c))

    def get(self, name, n=None):
        buf = np.zeros(n if n is not None else 1024 * 18, dtype=np.float64)
        k = self.L.gqo_get(self.h, name.encode(), _p(buf), buf.size)
        if k < 0:
            raise KeyError(self.L.gqo_last_error().decode())
        return buf[:k].copy()

    def __getattr__(self, name):  # o.qpos, o.M ...
        if name.startswith('_') or name in ('mm', 'L', 'h', 'nv', 'nq', 'nu'):
            raise AttributeError(name)
        v = self.get(name)
        if name == 'M':
            return v.reshape(self.nv, self.nv)
        if name == 'efc_J':
            return v.reshape(-1, self.nv)
        if name in ('xpos', 'xipos', 'subtree_com', 'geom_xpos', 'contact_pos'):
            return v.reshape(-1, 3)
        if name in ('xmat', 'geom_xmat', 'contact_frame'):
            return v.reshape(-1, 3, 3)
        if name in ('xquat',):
            return v.reshape(-1, 4)
        if name in ('cvel', 'contact_force'):
            return v.reshape(-1, 6)
        if name in ('nefc', 'ncon', 'solver_niter', 'warning'):
            return int(v[0])
        if name == 'time':
            return float(v[0])
        return v

    def jac(self, point, body):
        jp, jr = np.zeros((3, self.nv)), np.zeros((3, self.nv))
        pt = np.ascontiguousarray(point, dtype=np.float64)
        self.L.gqo_jac_point(self.h, _p(pt), int(body), _p(jp), _p(jr))
        return jp, jr

    def get_obs(self, obs_names, cmd=(0, 0, 0, 0), legs_order=(0, 1, 2, 3)):
        ids = np.asarray(obs_ids_from_names(obs_names), dtype=np.int32)
        dim = sum(OBS_DIMS[i] for i in ids)
        out = np.zeros(dim)
        cmd = np.asarray(cmd, dtype=np.float64)
        lo = np.asarray(legs_order, dtype=np.int32)
        term, inv = C.c_int(0), C.c_int(0)
        n = self.L.gqo_get_obs(self.h, _p(cmd), _p(lo), _p(ids), len(ids), _p(out), C.byref(term), C.byref(inv))
        if n != dim:
            raise RuntimeError(f'gqo_get_obs returned {n}, expected {dim}')
        res, k = {}, 0
        for name, i in zip(obs_names, ids):
            res[name] = out[k:k + OBS_DIMS[i]].copy()
            k += OBS_DIMS[i]
        return res, bool(term.value), bool(inv.value)

    def rollout(self, ctrl_seq, obs_names, cmd=(0, 0, 0, 0), legs_order=(0, 1, 2, 3), reset_on_term=True):
        ids = np.asarray(obs_ids_from_names(obs_names), dtype=np.int32)
        seq = np.ascontiguousarray(ctrl_seq, dtype=np.float64)
        cmd = np.asarray(cmd, dtype=np.float64)
        lo = np.asarray(legs_order, dtype=np.int32)
        return self.L.gqo_rollout(self.h, _p(seq), seq.shape[0], _p(cmd), _p(lo), _p(ids), len(ids), None, int(reset_on_term))
