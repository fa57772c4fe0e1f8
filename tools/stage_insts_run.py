"""Workload of tools/stage_insts.sh: a warmed-up 4096-env rollout state is built with the FULL kernel in a first
process-independent way (seeded), then 20 steps run under whatever GQ_STOP_STAGE the environment sets."""
import os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
stop = os.environ.pop('GQ_STOP_STAGE', '0')
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
n = 4096
kw = dict(state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1000,
          self_collision=False if os.environ.get('GQ_STAGE_NOSELF') else None)
full = QuadrupedEnv('mini_cheetah', **kw)                 # GQ_STOP_STAGE unset: complete steps build the state
full.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
pool = [torch.randn(n, 12, generator=g, device='cuda') * 50 for _ in range(16)]
for i in range(150):
    full.step(pool[i % 16])
os.environ['GQ_STOP_STAGE'] = stop
cut = QuadrupedEnv('mini_cheetah', **kw)                  # reads GQ_STOP_STAGE at batch creation
cut.reset(random=True)
cut.load_state_dict(full.state_dict())
torch.cuda.synchronize()
for i in range(20):
    cut.step(pool[i % 16])
    if stop != '0':
        cut.load_state_dict(full.state_dict())            # a cut step writes nothing back; keep the state fixed anyway
torch.cuda.synchronize()
