"""Histogram of Newton iterations (and of the reason the loop ended) over a 4096-env rollout state, per robot.

    python tools/niter_hist.py go2 hyqreal1 mini_cheetah

Reads the kernel's inspection record on the device (gq_debug_device_buffer / gq_debug_field).  Exit codes (gq_newton.h):
1 small predicted improvement, 2 iteration cap, 3 gradient tolerance, 4 fp32 noise floor, 5 not a descent direction,
6 converged step (no row changed piece)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd import _lib  # noqa: E402
from gym_quadruped_amd.accessors import _DevPtr  # noqa: E402
from gym_quadruped_amd.quadruped_env import QuadrupedEnv  # noqa: E402

for robot in sys.argv[1:] or ['mini_cheetah']:
    n = 4096
    env = QuadrupedEnv(robot, state_obs_names=('qpos',), num_envs=n, auto_reset='next_step', seed=1000)
    env.reset(random=True)
    g = torch.Generator(device='cuda').manual_seed(0)
    for i in range(200):
        env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
    env.enable_debug(n)
    env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
    torch.cuda.synchronize()
    ptr, nn, st = C.c_void_p(), C.c_int32(), C.c_int32()
    _lib.check(env._L.gq_debug_device_buffer(env._hbatch, C.byref(ptr), C.byref(nn), C.byref(st)), 'gq_debug_device_buffer')
    T = torch.as_tensor(_DevPtr(ptr.value, (nn.value, st.value)), device='cuda')

    def field(name):
        off, cnt = C.c_int32(), C.c_int32()
        _lib.check(env._L.gq_debug_field(name.encode(), C.byref(off), C.byref(cnt)), 'gq_debug_field')
        return T[:, off.value:off.value + cnt.value].cpu().numpy()

    nit, nefc, ex = field('niter')[:, 0], field('nefc')[:, 0], field('timer')[:, 23]
    print(f'{robot}: niter histogram {np.bincount(np.minimum(nit.astype(int), 30)).tolist()} max {nit.max():.0f}; nefc mean {nefc.mean():.1f} max {nefc.max():.0f}')
    print(f'   exit codes {np.bincount(ex.astype(int), minlength=7).tolist()}')
    for k in range(1, int(nit.max()) + 1):
        sel = nit.astype(int) == k
        if sel.sum(): print(f'   niter {k}: {int(sel.sum()):5d} envs, exit codes {np.bincount(ex[sel].astype(int), minlength=7).tolist()}')
