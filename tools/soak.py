"""Long random-action soak through the persistent rollout: every robot x scene, <steps> steps of 4096 envs in chunks, checking
that state and observations stay finite, envs keep being re-spawned, nobody sits outside the terrain limits or below the
floor, and no reset was flagged lift_failed.   python tools/soak.py [steps] [robot ...]"""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
robots = sys.argv[2:] or ['mini_cheetah', 'aliengo', 'b2', 'hyqreal2', 'go1', 'go2', 'hyqreal1', 'spot']
n, K = 4096, 500
bad = 0
for robot in robots:
    for scene in ('flat', 'random_boxes', 'perlin', 'stairs', 'slippery'):
        env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=7)
        env.reset(random=True)
        g = torch.Generator(device='cuda').manual_seed(1)
        t0 = time.perf_counter(); ok = True; zmin = 1e9
        for c in range(steps // K):
            acts = torch.randn(K, n, 12, generator=g, device='cuda') * 50
            env.rollout(acts, shards=0)
            fin = bool(torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all() and torch.isfinite(env._obs_buf).all())
            zmin = min(zmin, float(env.qpos[:, 2].min()))
            if not fin: ok = False; break
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        lim = env.terrain_limits; q = env.qpos
        inside = bool(((q[:, 0] <= lim[0] + 1) & (q[:, 0] >= lim[1] - 1) & (q[:, 1] <= lim[2] + 1) & (q[:, 1] >= lim[3] - 1)).all())
        lf = int(env.lift_failed.sum())
        status = 'ok' if (ok and inside and lf == 0 and zmin > -0.5) else 'FAIL'
        bad += status != 'ok'
        print(f'{robot:13s} {scene:13s} {status}: {steps} steps, {n * (steps // K) * K / dt / 1e6:6.1f} M env-steps/s (incl. drawing the actions), episodes max {int(env._episode.max())}, min base z {zmin:.3f}, lift_failed {lf}, finite {ok}, inside {inside}', flush=True)
        del env
print('soak', 'PASSED' if bad == 0 else f'FAILED ({bad})')
