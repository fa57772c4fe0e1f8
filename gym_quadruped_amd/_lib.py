"""Loader of the HIP extension ``libgq.so`` (C-ABI of include/gq.h).  There is NO fallback: if the library is
missing or cannot be loaded the product path raises - it never routes through the CPU oracle."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from .cabi import GQ_ABI_VERSION, GqImuCfg, GqMailboxView, GqModelDesc, GqObsOut, GqPolicyPd, GqResampleCfg, GqResetCfg, GqState

_LIB = None
import os

# GQ_LIBGQ_PATH: developer override used for A/B timing of two kernel builds inside one GPU session
LIB_PATH = Path(os.environ.get('GQ_LIBGQ_PATH', Path(__file__).parent / 'libgq.so'))

EXPORTS = ['gq_last_error', 'gq_version', 'gq_struct_sizes', 'gq_obs_dim', 'gq_model_create', 'gq_model_destroy', 'gq_batch_create',
           'gq_batch_destroy', 'gq_batch_obs_dim', 'gq_batch_set_imu', 'gq_batch_set_heightmap', 'gq_batch_set_pair_exchange', 'gq_batch_set_pending', 'gq_batch_set_resampling', 'gq_heightmap', 'gq_heightmap_strided', 'gq_step_range', 'gq_batch_bind', 'gq_rollout', 'gq_rollout_closed', 'gq_rollout_closed_status', 'gq_mailbox_get', 'gq_jac', 'gq_ray', 'gq_forward', 'gq_full_mass', 'gq_batch_set_outputs', 'gq_contact_force', 'gq_step', 'gq_reset', 'gq_debug_enable', 'gq_debug_get', 'gq_debug_device_buffer', 'gq_debug_field', 'gq_debug_stop_stage']


class GqError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        raise GqError(f'{LIB_PATH} not found: build the HIP extension first (python -c "import __graft_entry__ as g; '
                      f'g.build()" or make -C gym_quadruped_amd/csrc). There is no CPU fallback.')
    import torch  # noqa: F401  - load torch's HIP runtime first so libgq.so binds to the same one
    L = C.CDLL(str(LIB_PATH))
    L.gq_last_error.restype = C.c_char_p
    # a stale libgq.so (built from an older include/gq.h) would shift every by-value struct argument: refuse it here
    if not hasattr(L, 'gq_struct_sizes') or L.gq_version() != GQ_ABI_VERSION:
        raise GqError(f'{LIB_PATH} implements ABI {L.gq_version()}, this binding expects {GQ_ABI_VERSION}: rebuild the library '
                      f'(make -C gym_quadruped_amd/csrc)')
    sizes = (C.c_int32 * 8)()
    L.gq_struct_sizes(sizes)
    mirror = [C.sizeof(t) for t in (GqModelDesc, GqState, GqObsOut, GqResetCfg, GqResampleCfg, GqImuCfg, GqPolicyPd, GqMailboxView)]
    if list(sizes) != mirror:
        raise GqError(f'struct layouts differ between {LIB_PATH} {list(sizes)} and gym_quadruped_amd/cabi.py {mirror}')
    L.gq_model_create.argtypes = [C.POINTER(GqModelDesc), C.c_int, C.POINTER(C.c_void_p)]
    L.gq_model_destroy.argtypes = [C.c_void_p]
    L.gq_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.gq_batch_destroy.argtypes = [C.c_void_p]
    L.gq_batch_obs_dim.argtypes = [C.c_void_p]
    L.gq_batch_set_imu.argtypes = [C.c_void_p, C.POINTER(GqImuCfg), C.c_void_p]
    L.gq_batch_set_heightmap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
    L.gq_batch_set_pair_exchange.argtypes = [C.c_void_p, C.c_int]
    L.gq_batch_set_resampling.argtypes = [C.c_void_p, C.POINTER(GqResampleCfg), C.POINTER(GqResetCfg), C.c_void_p, C.c_void_p]
    L.gq_batch_set_pending.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, GqState, GqObsOut, C.POINTER(GqResetCfg), C.c_void_p,
                          C.c_void_p, C.c_void_p]
    L.gq_step_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, GqState, GqObsOut, C.POINTER(GqResetCfg), C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, GqState, GqObsOut, C.POINTER(GqResetCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_rollout_closed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(GqPolicyPd), C.c_int, C.c_int, C.c_double, GqState, GqObsOut, C.POINTER(GqResetCfg), C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_rollout_closed_status.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
    L.gq_mailbox_get.argtypes = [C.c_void_p, C.POINTER(GqMailboxView)]
    L.gq_batch_bind.argtypes = [C.c_void_p, GqState, GqObsOut, C.POINTER(GqResetCfg), C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(GqResetCfg), GqState, GqObsOut,
                           C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_jac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, GqState, GqObsOut, C.c_void_p]
    L.gq_heightmap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.gq_heightmap_strided.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.gq_debug_enable.argtypes = [C.c_void_p, C.c_int]
    L.gq_debug_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int]
    L.gq_obs_dim.argtypes = [C.c_int]
    L.gq_full_mass.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.gq_batch_set_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.gq_contact_force.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.gq_debug_device_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.gq_debug_stop_stage.argtypes = [C.c_void_p, C.c_int]
    L.gq_debug_field.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    _LIB = L
    return L


def check(rc, what):
    if rc < 0:
        raise GqError(f'{what} failed ({rc}): {lib().gq_last_error().decode()}')
    return rc
