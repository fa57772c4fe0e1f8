"""Cost of a batched reset (reset kernel + the reset's own step) for a robot / scene: python tools/reset_probe.py [robot] [scene] [n]
With GQ_LIBGQ_PATH pointing at development builds with -DGQ_LIFT_CAP=<k> this separates the lift loop's iterations."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
robot = sys.argv[1] if len(sys.argv) > 1 else 'hyqreal1'
scene = sys.argv[2] if len(sys.argv) > 2 else 'random_boxes'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
env = QuadrupedEnv(robot, scene=scene, state_obs_names=('qpos', 'qvel'), num_envs=n, auto_reset='next_step', seed=3)
for _ in range(5): env.reset(random=True)
torch.cuda.synchronize()
ts = []
for _ in range(40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); env.reset(random=True); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
print(f'{robot} {scene} {n} envs: reset() = reset kernel + its step: median {np.median(ts):.1f} us, min {min(ts):.1f}, max {max(ts):.1f}; lift_failed {int(env.lift_failed.sum())}')

if __import__('os').environ.get('GQ_RESET_STAMPS'):   # library built with -DGQ_RESET_STAMPS: per-env stamps in qfrc_applied[6:11] (scaled 1e-9)
    S = env._applied[:, 6:11].cpu().numpy().astype(np.float64) * 1e9
    lab = ['draws + kinematics + scan + spheres (cycles)', 'lift loop (cycles)', 'lift iterations', 'box scans issued', 'candidate boxes (max per iteration)']
    for k, l in enumerate(lab):
        x = S[:, k]
        print(f'  {l:48s} mean {x.mean():10.1f}  p50 {np.median(x):10.1f}  p95 {np.percentile(x, 95):10.1f}  max {x.max():10.1f}')
    w = int(np.argmax(S[:, 1])); print('  worst env', w, S[w].round(0).tolist())
