"""Edge audit of the batched API on awkward sizes and option mixes: every rollout form against the plain step loop (bit for bit),
small and non-power-of-two batches, both auto-reset modes, solvers, scenes, masks.   usage: api_edge_fuzz.py"""
import itertools, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
OBS = ('qpos', 'qvel', 'contact_state', 'contact_forces', 'base_lin_vel')
F = ('_qpos', '_qvel', '_qacc', '_warm', '_time', '_step_num', '_episode', '_terminated', '_truncated', '_obs_buf')
bad = 0
def twin(robot, n, scene, solver, ar, **kw):
    mk = lambda: QuadrupedEnv(robot, scene=scene, state_obs_names=OBS, num_envs=n, auto_reset=ar, solver=solver, seed=7, **kw)
    a, b = mk(), mk()
    a.reset(random=True); b.reset(random=True)
    return a, b
def same(a, b):
    torch.cuda.synchronize()
    return [f for f in F if not torch.equal(getattr(a, f), getattr(b, f))]
cases = [('mini_cheetah', 'flat', 'newton'), ('mini_cheetah', 'flat', 'pgs'), ('aliengo', 'perlin', 'newton'), ('aliengo', 'random_boxes', 'pgs'), ('go2', 'flat', 'newton')]
g = torch.Generator(device='cuda').manual_seed(0)
for (robot, scene, solver), n, ar in itertools.product(cases, (1, 3, 64, 130), ('next_step', 'same_step', False)):
    kw = dict(self_collision=True) if solver == 'pgs' and robot == 'aliengo' else {}
    a, b = twin(robot, n, scene, solver, ar, **kw)
    for K, shards in ((1, 0), (4, 0), (1, 1), (3, 2), (5, 3)):
        acts = torch.randn(K, n, 12, generator=g, device='cuda') * 60
        if ar == 'same_step' and shards == 0: continue   # refused by design: the persistent rollout needs next-step auto-reset or none
        try:
            b.rollout(acts, shards=shards)
        except Exception as e:
            print(robot, scene, solver, n, ar, K, shards, 'ERROR', str(e)[:140], flush=True); bad += 1; break
        for k in range(K): a.step(acts[k])
        d = same(a, b)
        if d:
            print(robot, scene, solver, n, ar, K, shards, 'DIFFERENT', d, flush=True); bad += 1; break
    else:
        # masked reset leaves the others alone and keeps the twins in step
        ids = list(range(0, n, 2))
        a.reset(random=True, env_ids=ids); b.reset(random=True, env_ids=ids)
        act = torch.randn(n, 12, generator=g, device='cuda') * 60
        a.step(act); b.step(act)
        d = same(a, b)
        if d: print(robot, scene, solver, n, ar, 'masked reset DIFFERENT', d, flush=True); bad += 1
    a.close(); b.close()
print('cases with a failure:', bad)
sys.exit(1 if bad else 0)
