"""A policy IN the loop without launch boundaries: the reference's control loop

    obs = env.reset()
    while True:
        action = policy(obs)                 # here: joint-space PD towards the standing posture (+ exploration noise)
        obs, reward, terminated, truncated, info = env.step(action)        # quadruped_env.py:251-307, README.md:31-33

played on the device for a whole batch (``QuadrupedEnv.rollout_closed_loop`` -> ``gq_rollout_closed``): env e's step k + 1 waits
for env e's action only, never for the slowest env of step k.  Two placements of the policy:

* ``mode='inline'``  - the wavefront that steps an env evaluates the policy on the observation row it has just written;
* ``mode='mailbox'`` - the policy is a kernel of its own on a second HIP stream; observations and actions travel through per-env
  mailboxes, ready envs through per-XCD queues, and the wavefronts of ONE persistent step launch pop env-steps as tasks (any
  resident kernel that follows the protocol of include/gq.h can take the policy's seat).

Both end in the state of the plain step loop fed with the same actions, bit for bit - shown below.

    python examples/closed_loop_pd.py [n_envs] [steps]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
KP, KD = 25.0, 0.8
mk = lambda: QuadrupedEnv('mini_cheetah', state_obs_names=('qpos_js', 'qvel_js', 'base_lin_vel', 'base_ori_euler_xyz', 'contact_state'), num_envs=n,
                          device='cuda:0', auto_reset='next_step', seed=3)

# 1. the closed loop the reference's way: one launch per step, the policy as torch code between the launches
env = mk()
obs = env.reset(random=True)
q_des = env._key_qpos[7:19].float()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    action = KP * (q_des - obs['qpos_js']) - KD * obs['qvel_js']
    obs, reward, terminated, truncated, info = env.step(action)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'step loop with a torch policy between the launches: {n * K / dt / 1e6:7.2f} M env-steps/s  ({dt / K * 1e6:.1f} us per step)')
ref = env.qpos.clone()

# 2. the same loop on the device
for mode in ('inline', 'mailbox'):
    e2 = mk()
    e2.reset(random=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = e2.rollout_closed_loop(K, KP, KD, mode=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = bool(torch.equal(e2.qpos, ref))
    print(f'closed-loop persistent rollout, {mode:7s} policy:        {n * K / dt / 1e6:7.2f} M env-steps/s  ({dt / K * 1e6:.1f} us per step); '
          f'final state equal to the step loop bit for bit: {same}; standing envs: {int((out["obs"]["contact_state"].sum(1) >= 3).sum())} of {n}')
    e2.close()

# 3. an exploring policy (the PD law + Gaussian torque noise): robots fall and re-spawn, the regime of the benchmark's random actions
e3 = mk()
e3.reset(random=True)
e3.rollout_closed_loop(200, KP, KD, noise_sigma=50.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
e3.rollout_closed_loop(K, KP, KD, noise_sigma=50.0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'closed loop, inline PD + N(0, 50) torque noise:            {n * K / dt / 1e6:7.2f} M env-steps/s  ({dt / K * 1e6:.1f} us per step); episodes so far: max {int(e3._episode.max())}')
