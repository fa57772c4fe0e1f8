"""Closed-loop rollouts against the step loop, at scale: for every robot x scene of the list, a mailbox-mode and an inline-mode closed-loop
rollout (PD + exploration noise, so that envs fall and re-spawn) is replayed through the plain step loop with the recorded actions and
compared BIT FOR BIT, step by step - millions of env-steps in which an env changes wavefronts (CUs) every step.  Any stale read of a
per-env word (the L1 / scalar-cache hazards of DESIGN.md section 2) shows up as a first-differing (env, step).
    python tools/closed_loop_soak.py [n_envs] [K] [trials] [robot:scene ...]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cases = [c.split(':') for c in sys.argv[4:]] or [['mini_cheetah', 'flat'], ['aliengo', 'flat'], ['go2', 'flat'], ['hyqreal1', 'flat'], ['spot', 'flat'], ['b2', 'flat'], ['go1', 'flat'],
                                                  ['hyqreal2', 'flat'], ['aliengo', 'perlin'], ['hyqreal1', 'random_boxes'], ['go2', 'stairs'], ['mini_cheetah', 'random_pyramids']]
names = ('qpos_js', 'qvel_js', 'tau_ctrl_setpoint', 'base_lin_vel', 'contact_forces', 'contact_state')
STATE = ('_qpos', '_qvel', '_qacc', '_warm', '_time', '_step_num', '_episode', '_cmd', '_terminated', '_truncated', '_invalid', '_obs_buf', '_friction', '_contacts_dropped', '_h9')
total = bad = 0
t_all = time.perf_counter()
for robot, scene in cases:
    for mode in ('mailbox', 'inline'):
        nbad = nsteps = 0
        for trial in range(trials):
            mk = lambda: QuadrupedEnv(robot, scene=scene, num_envs=n, device='cuda:0', solver='newton', auto_reset='next_step', seed=100 + trial, state_obs_names=names,
                                      base_vel_command_type='random+reset')
            a, b = mk(), mk()
            a.reset(random=True); b.reset(random=True)
            a._h9[:, 1] = 30; b._h9[:, 1] = 30      # command redraws inside the rollout
            r = b.rollout_closed_loop(K, 25.0, 0.8, mode=mode, noise_sigma=30.0, record_obs=True, record_actions=True)
            acts = r['actions']
            first = {}
            for k in range(K):
                a.step(acts[k])
                diff = (a._obs_buf != r['obs_seq'][k]).any(dim=1)
                if bool(diff.any()):
                    for e in diff.nonzero().flatten().tolist():
                        first.setdefault(e, k)
            torch.cuda.synchronize()
            same = all(torch.equal(getattr(a, k), getattr(b, k)) for k in STATE)
            nbad += len(first) + (0 if same or first else 1)
            nsteps += n * K
            if first:
                print(f'   {robot} {scene} {mode} trial {trial}: first differences (env: step) {sorted(first.items())[:8]}', flush=True)
            a.close(); b.close()
        total += nsteps; bad += nbad
        print(f'{robot:13s} {scene:16s} {mode:8s}: {nsteps:9d} env-steps replayed, {nbad} envs left the step loop, episodes max {int(b._episode.max()) if False else ""}', flush=True)
print(f'closed-loop soak: {total} env-steps, {bad} mismatching envs, {time.perf_counter() - t_all:.0f} s ->', 'PASSED' if bad == 0 else 'FAILED')
