"""Throughput of the persistent open-loop rollout (gq_rollout shards = 0) against the step loop on any robot / scene.
    python tools/persistent_probe.py <robot> [scene] [K]"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
robot = sys.argv[1] if len(sys.argv) > 1 else 'go2'
scene = sys.argv[2] if len(sys.argv) > 2 else 'flat'
K = int(sys.argv[3]) if len(sys.argv) > 3 else 250
n = 4096
env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1000)
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
acts = torch.randn(K, n, 12, generator=g, device='cuda') * 50
for k in range(300): env.step(acts[k % K])
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(K): env.step(acts[k])
torch.cuda.synchronize(); t1 = time.perf_counter()
env.rollout(acts, shards=0); torch.cuda.synchronize(); t2 = time.perf_counter()
env.rollout(acts, shards=0); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f'{robot} {scene}: step loop {n * K / (t1 - t0) / 1e6:.2f} M env-steps/s, persistent rollout {n * K / (t3 - t2) / 1e6:.2f} M ({K} steps, 4096 envs)')
