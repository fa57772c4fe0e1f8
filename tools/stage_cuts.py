"""Critical path by stage on the PRODUCTION step kernel, in steady state.

A launch lasts as long as its slowest wave.  Every iteration launches the kernel twice on the same state: once cut short
after stage marker k (gq_debug_stop_stage; markers <= 10 write nothing) and timed with HIP events, once complete (untimed)
so that the rollout keeps evolving and caches stay as warm as in the benchmark.  time(cut k) - time(cut k-1) is what
stage k adds to the critical path of the batch.

    python tools/stage_cuts.py [n_envs] [robot]
"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
from gym_quadruped_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
robot = sys.argv[2] if len(sys.argv) > 2 else 'mini_cheetah'
scene = sys.argv[3] if len(sys.argv) > 3 else 'flat'
env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1000)
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
pool = [torch.randn(n, 12, generator=g, device='cuda') * 50 for _ in range(16)]
for i in range(300):
    env.step(pool[i % 16])
names = {15: 'launch floor (nothing done)', 1: 'S0+S1 load+kinematics', 2: 'S2 inertias', 3: 'S3 mass matrix', 4: 'S4 factor x2', 5: 'S5 rne+actuation', 14: 'S6a collision scan',
         6: 'S6b contact list', 7: 'S7 rows', 8: 'S8 qacc_smooth', 9: 'S9 solver', 10: 'S10 accelerations', 0: 'euler + S11 obs + gather (full)'}
stop = lambda k: _lib.check(env._L.gq_debug_stop_stage(env._hbatch, k), 'gq_debug_stop_stage')
prev = 0.0
print(f'{robot}, {n} envs: launch time of the kernel cut after each stage (us, mean of 300 launches interleaved with complete steps)')
for k in [15, 1, 2, 3, 4, 5, 14, 6, 7, 8, 9, 10, 0]:
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(300)]
    torch.cuda.synchronize()
    for i in range(300):
        if k:
            stop(k); ev[i][0].record(); env.step(pool[i % 16]); ev[i][1].record(); stop(0)
            env.step(pool[i % 16])
        else:
            env.step(pool[(i + 5) % 16])
            ev[i][0].record(); env.step(pool[i % 16]); ev[i][1].record()
    torch.cuda.synchronize()
    t = float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e3
    print(f'  {names[k]:34s} {t:8.1f}   +{t - prev:7.1f}')
    prev = t
