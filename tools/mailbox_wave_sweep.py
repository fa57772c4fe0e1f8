"""Mailbox-mode closed loop: throughput against the number of step workgroups (an env is not bound to a wavefront, so fewer wavefronts than envs
keep stepping while the others' actions are in flight).   usage: mailbox_wave_sweep.py [n_envs] [K] [sigma]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sig = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
env = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
env.reset(random=True)
env.rollout_closed_loop(300, 25.0, 0.8, mode='inline', noise_sigma=sig)
sd = env.state_dict(); la = env._launches
def run(**kw):
    env.load_state_dict(sd); env._launches = la
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout_closed_loop(K, 25.0, 0.8, noise_sigma=sig, **kw)
    torch.cuda.synchronize(); return n * K / (time.perf_counter() - t0) / 1e6
print(f'{n} envs, K={K}, sigma={sig}: inline {run(mode="inline"):.1f} M')
for pw in (64, 128):
    for frac in (1.0, 0.9375, 0.875, 0.8125, 0.75, 0.625, 0.5):
        sw = int(min(n, 4096) * frac)
        print(f'  mailbox, policy waves {pw:3d}, step workgroups {sw:5d}: {run(mode="mailbox", policy_waves=pw, step_waves=sw):6.1f} M', flush=True)
