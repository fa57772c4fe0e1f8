"""Does QuadrupedEnv.rollout overlap its shard launches?  Run under rocprofv3 --kernel-trace and inspect start/end times."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
n = 4096
shards = int(sys.argv[1]) if len(sys.argv) > 1 else 4
env = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
acts = torch.randn(64, n, 12, generator=g, device='cuda') * 50
for _ in range(3): env.rollout(acts, shards=shards)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): env.rollout(acts, shards=shards)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'shards {shards}: {n * 256 / dt / 1e6:.2f} M env-steps/s, {dt / 256 * 1e6:.1f} us/step')
