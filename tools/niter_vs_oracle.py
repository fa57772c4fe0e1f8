"""Newton iterations of the kernel (fp32) against the oracle (fp64, the reference's algorithm) on the SAME rollout states.

    python tools/niter_vs_oracle.py [robot] [n_sample] [scene]

Takes the 4096-env benchmark rollout after 300 random-action steps, steps a sample of envs once more on the GPU (inspection
record: niter, exit code) and from the same states in the oracle, and prints the joint histogram.  Says whether the long tail
of iteration counts (which sets the launch time: a launch lasts as long as its slowest wave) belongs to the algorithm or to the
kernel's fp32 arithmetic.  Test-side tool: the oracle is the checker here, never the product."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv  # noqa: E402
from helpers import marshalled  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else 'go2'
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 384
scene = sys.argv[3] if len(sys.argv) > 3 else 'flat'
n = 4096
env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1000)
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
for i in range(300): env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
st = {k.lstrip('_'): (v.clone().cpu().numpy() if torch.is_tensor(v) else v) for k, v in env.state_dict().items()}
act = torch.randn(n, 12, generator=g, device='cuda') * 50
env.enable_debug(n)
env.step(act); torch.cuda.synchronize()
d = env.debug_internals(n, ['niter', 'nefc', 'timer'])
knit = np.array([x['niter'][0] for x in d]).astype(int); kex = np.array([x['timer'][23] for x in d]).astype(int); knefc = np.array([x['nefc'][0] for x in d]).astype(int)
pend = st['terminated'].astype(bool)
mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-8, boxes=env.scene_desc.get('boxes') or None, hfield=env.scene_desc.get('hfield'), terrain_limits=env.terrain_limits)
o = Oracle(mm)
a = act.cpu().numpy().astype(np.float64)
idx = [e for e in np.argsort(-knit) if not pend[e]][:ns // 2] + [e for e in range(n) if not pend[e]][:ns // 2]
onit = []
onefc = []
for e in idx:
    o.set_state(st['qpos'][e], st['qvel'][e].astype(np.float64), st['warm'][e].astype(np.float64), st['applied'][e].astype(np.float64), float(st['time'][e]), float(st['friction'][e]))
    o.step(a[e])
    onit.append(int(o.solver_niter)); onefc.append(int(o.nefc))
onit = np.array(onit); kn = knit[idx]
onefc = np.array(onefc)
same = onefc == knefc[idx]
ratio = kn / np.maximum(onit, 1)
worst = np.argsort(-np.where(same, ratio, 0))[:6]
print(f'{robot} {scene}: same row count in {int(same.sum())} of {len(idx)}; among those kernel/oracle iteration ratio max {np.where(same, ratio, 0).max():.1f}; worst (kernel, oracle, nefc): ' + ', '.join(f'({kn[w]}, {onit[w]}, {onefc[w]})' for w in worst))
print(f'{robot}: {len(idx)} envs (half = the kernel\'s slowest, half = the first envs)')
print('kernel niter  mean %.2f max %d | oracle niter mean %.2f max %d | nefc kernel mean %.1f' % (kn.mean(), kn.max(), onit.mean(), onit.max(), knefc[idx].mean()))
H = np.zeros((16, 16), int)
for k_, o_ in zip(np.minimum(kn, 15), np.minimum(onit, 15)): H[k_, o_] += 1
print('rows: kernel niter, columns: oracle niter')
for r in range(16):
    if H[r].sum(): print(f'  {r:2d}: ' + ' '.join(f'{c:4d}' for c in H[r]))
print('kernel exit codes of the slowest half:', np.bincount(kex[idx[:ns // 2]], minlength=7).tolist())
