import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
n = 4096
env = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
env.reset()
g = torch.Generator(device='cuda').manual_seed(0)
for i in range(60): env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
env.enable_debug(n)
N = []; E = []
for i in range(6):
    env.step(torch.randn(n, 12, generator=g, device='cuda') * 50); torch.cuda.synchronize()
    d = env.debug_internals(n, ['niter', 'nefc'])
    N.append(np.array([x['niter'][0] for x in d]).astype(int)); E.append(np.array([x['nefc'][0] for x in d]).astype(int))
for i in range(1, 6):
    a, b = N[i - 1], N[i]
    print('step', i, 'corr niter(t-1),niter(t)', np.corrcoef(a, b)[0, 1].round(3), ' corr nefc,niter', np.corrcoef(E[i], b)[0, 1].round(3), 'corr |dnefc|,niter', np.corrcoef(np.abs(E[i] - E[i - 1]), b)[0, 1].round(3))
    heavy = b >= 4
    print('   heavy now (>=4):', heavy.sum(), ' of which prev>=2:', (a[heavy] >= 2).sum(), 'prev>=3:', (a[heavy] >= 3).sum(), '; waves with prev>=2:', (a >= 2).sum(), 'prev>=3:', (a >= 3).sum())
    print('   heavy by nefc change: dnefc!=0 among heavy', (E[i][heavy] != E[i-1][heavy]).sum(), ' overall dnefc!=0:', (E[i] != E[i-1]).sum())
