"""Throughput of the closed-loop persistent rollout (gq_rollout_closed) with a PD policy in the loop, next to the step loop and the
open-loop persistent rollout fed with the SAME actions (a standing robot is a different workload from a randomly actuated one).
usage: closed_loop_probe.py [robot] [n_envs] [K] [scene]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

robot = sys.argv[1] if len(sys.argv) > 1 else 'mini_cheetah'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 500
scene = sys.argv[4] if len(sys.argv) > 4 else 'flat'
KP, KD = 25.0, 0.8
mk = lambda: QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
env = mk()
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
pool = [torch.randn(n, 12, generator=g, device='cuda') * 50 for _ in range(16)]
for i in range(300): env.step(pool[i % 16])
torch.cuda.synchronize()

def timed(fn, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n * steps / dt / 1e6, dt / steps * 1e6

m, us = timed(lambda: [env.step(pool[i % 16]) for i in range(K)], K)
print(f'{robot} {scene} n={n}: step loop, random actions             {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
acts = torch.stack([pool[i % 16] for i in range(K)])
m, us = timed(lambda: env.rollout(acts, shards=0), K)
print(f'{robot} {scene} n={n}: persistent open loop, random actions  {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
# settle into the closed-loop regime, checkpoint, record the policy's actions from there
env.rollout_closed_loop(300, KP, KD, mode='inline')
sd = env.state_dict()
rec = env.rollout_closed_loop(K, KP, KD, mode='inline', record_actions=True)['actions']
def restore():
    env.load_state_dict(sd); torch.cuda.synchronize()
restore(); m, us = timed(lambda: [env.step(rec[i]) for i in range(K)], K)
print(f'{robot} {scene} n={n}: step loop, the PD policy\'s actions    {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
restore(); m, us = timed(lambda: env.rollout(rec, shards=0), K)
print(f'{robot} {scene} n={n}: persistent open loop, same actions    {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
restore(); m, us = timed(lambda: env.rollout_closed_loop(K, KP, KD, mode='inline'), K)
print(f'{robot} {scene} n={n}: CLOSED loop, inline PD policy         {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
for pw, sw in ((64, 0), (256, 0), (64, n - 256 if n > 512 else 0), (16, 0)):
    try:
        restore(); m, us = timed(lambda: env.rollout_closed_loop(K, KP, KD, mode='mailbox', policy_waves=pw, step_waves=sw), K)
        print(f'{robot} {scene} n={n}: CLOSED loop, mailbox, policy waves {pw:3d} step waves {sw or n:5d}: {m:7.2f} M env-steps/s  {us:6.1f} us/step  status {env.closed_loop_status()}', flush=True)
    except Exception as ex:
        print('closed loop failed:', pw, sw, ex, flush=True)
        break
# an exploring policy: the same PD law + Gaussian torque noise of the benchmark's amplitude (robots fall and re-spawn like under random actions)
SIG = 50.0
env.rollout_closed_loop(300, KP, KD, mode='inline', noise_sigma=SIG)
sd = env.state_dict(); launches = env._launches
rec = env.rollout_closed_loop(K, KP, KD, mode='inline', noise_sigma=SIG, record_actions=True)['actions']
def restore():
    env.load_state_dict(sd); env._launches = launches; torch.cuda.synchronize()
restore(); m, us = timed(lambda: [env.step(rec[i]) for i in range(K)], K)
print(f'{robot} {scene} n={n}: step loop, the noisy policy\'s actions {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
restore(); m, us = timed(lambda: env.rollout(rec, shards=0), K)
print(f'{robot} {scene} n={n}: persistent open loop, same actions    {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
restore(); m, us = timed(lambda: env.rollout_closed_loop(K, KP, KD, mode='inline', noise_sigma=SIG), K)
print(f'{robot} {scene} n={n}: CLOSED loop, inline PD + noise        {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
restore(); m, us = timed(lambda: env.rollout_closed_loop(K, KP, KD, mode='mailbox', noise_sigma=SIG), K)
print(f'{robot} {scene} n={n}: CLOSED loop, mailbox PD + noise       {m:7.2f} M env-steps/s  {us:6.1f} us/step  status {env.closed_loop_status()}', flush=True)
print(f'contacts per env (mean of contact_state sum) {float(env._obs_views["contact_state"].sum(1).mean()):.2f}, episodes max {int(env._episode.max())}')
