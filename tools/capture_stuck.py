"""Find envs whose Newton solve runs into the iteration cap on a benchmark-like rollout and save their PRE-step state
(qpos, qvel, warm start, applied force, friction, control) for replay under the emulator / oracle.
Usage: capture_stuck.py [robot] [n] [steps] [threshold] [scene]  ->  gpurun_out/stuck_<robot>.npz"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

robot = sys.argv[1] if len(sys.argv) > 1 else 'go1'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
thr = int(sys.argv[4]) if len(sys.argv) > 4 else 40
scene = sys.argv[5] if len(sys.argv) > 5 else 'flat'
env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
env.enable_debug(n)
names = ['_qpos', '_qvel', '_warm', '_applied', '_time', '_friction', '_cmd', '_step_num', '_terminated']
cap = {k: [] for k in names + ['ctrl', 'niter', 'nefc', 'step', 'env']}
for s in range(steps):
    a = torch.randn(n, 12, generator=g, device='cuda') * 50
    pre = {k: getattr(env, k).clone() for k in names}
    env.step(a); torch.cuda.synchronize()
    d = env.debug_internals(n, ['niter', 'nefc'])
    nit = np.array([x['niter'][0] for x in d]).astype(int)
    for e in np.where(nit >= thr)[0]:
        for k in names: cap[k].append(pre[k][e].cpu().numpy())
        cap['ctrl'].append(a[e].cpu().numpy()); cap['niter'].append(nit[e]); cap['nefc'].append(int(d[e]['nefc'][0])); cap['step'].append(s); cap['env'].append(e)
    hist = np.bincount(np.minimum(nit, 30), minlength=31) if s == 0 else hist + np.bincount(np.minimum(nit, 30), minlength=31)
    if s % 50 == 0: print(s, 'niter max', nit.max(), 'captured', len(cap['env']), flush=True)
    if len(cap['env']) >= 12: break
out = ROOT / 'gpurun_out' / f'stuck_{robot}_{scene}.npz'
np.savez(out, **{k: np.array(v) for k, v in cap.items()})
print('finite', bool(torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all()), 'niter histogram', hist.tolist())
print('captured', len(cap['env']), 'events ->', out, 'niter', cap['niter'], 'nefc', cap['nefc'], 'steps', cap['step'], 'envs', cap['env'])
