"""Throughput of the closed-loop persistent rollout (gq_rollout_closed: PD policy kernel on a second stream, per-XCD ready queues)
next to the step loop and the open-loop persistent rollout.   usage: closed_loop_probe.py [robot] [n_envs] [K] [scene]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

robot = sys.argv[1] if len(sys.argv) > 1 else 'mini_cheetah'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
scene = sys.argv[4] if len(sys.argv) > 4 else 'flat'
env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
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
print(f'{robot} {scene} n={n}: step loop (random actions)      {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
acts = torch.stack([pool[i % 16] for i in range(K)])
m, us = timed(lambda: env.rollout(acts, shards=0), K)
print(f'{robot} {scene} n={n}: persistent open-loop rollout    {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
env.rollout_closed_loop(50, 25.0, 0.8, mode='inline')
m, us = timed(lambda: env.rollout_closed_loop(K, 25.0, 0.8, mode='inline'), K)
print(f'{robot} {scene} n={n}: closed loop PD, inline policy   {m:7.2f} M env-steps/s  {us:6.1f} us/step', flush=True)
for pw, sw in ((64, 0), (128, 0), (256, 0), (64, n - 256 if n > 512 else 0), (16, 0)):
    try:
        env.rollout_closed_loop(50, 25.0, 0.8, mode='mailbox', policy_waves=pw, step_waves=sw)   # warm
        m, us = timed(lambda: env.rollout_closed_loop(K, 25.0, 0.8, mode='mailbox', policy_waves=pw, step_waves=sw), K)
        print(f'{robot} {scene} n={n}: closed loop PD, mailbox, policy waves {pw:3d} step waves {sw or n:5d}: {m:7.2f} M env-steps/s  {us:6.1f} us/step  status {env.closed_loop_status()}', flush=True)
    except Exception as ex:
        print('closed loop failed:', pw, sw, ex, flush=True)
        break
term = float(env._terminated.float().mean())
print(f'terminated fraction at the end {term:.4f}, episodes max {int(env._episode.max())}')
