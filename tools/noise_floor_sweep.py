"""Throughput and Newton iteration counts against the solver's fp32 noise floor (QuadrupedEnv(solver_noise_floor=...), include/gq.h GqModelDesc.noise_floor).
usage: noise_floor_sweep.py [robot ...]"""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

n, K = 4096, 1000
for robot in (sys.argv[1:] or ['go2', 'hyqreal1', 'mini_cheetah']):
    for nf in (0.0, 1e-5, 3e-5, 1e-4, 3e-4):
        env = QuadrupedEnv(robot, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1, solver_noise_floor=nf)
        env.reset(random=True)
        g = torch.Generator(device='cuda').manual_seed(0)
        pool = [torch.randn(n, 12, generator=g, device='cuda') * 50 for _ in range(16)]
        for i in range(300): env.step(pool[i % 16])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K): env.step(pool[i % 16])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        env.enable_debug(n)
        env.step(pool[0]); torch.cuda.synchronize()
        nit = np.array([d['niter'][0] for d in env.debug_internals(n, ['niter'])])
        print(f'{robot:13s} noise_floor {nf:7.0e}: {n * K / dt / 1e6:6.2f} M env-steps/s   niter mean {nit.mean():.2f} p99 {np.percentile(nit, 99):.0f} max {nit.max():.0f}', flush=True)
        env.close()
