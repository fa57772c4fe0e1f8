import sys, time, torch
sys.path.insert(0, '.')
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
n, K = 4096, 1500
env = QuadrupedEnv('mini_cheetah', state_obs_names=('qpos_js','qvel_js','base_lin_vel','contact_forces'), num_envs=n, auto_reset='next_step', seed=1)
env.reset(random=True)
env.rollout_closed_loop(300, 25.0, 0.8, noise_sigma=50.0, mode='inline')
for mode in ('mailbox', 'inline', 'mailbox', 'inline'):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout_closed_loop(K, 25.0, 0.8, noise_sigma=50.0, mode=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{mode:8s} {n*K/dt/1e6:7.2f} M  {dt/K*1e6:6.1f} us/step', flush=True)
