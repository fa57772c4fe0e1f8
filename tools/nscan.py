import sys, time, torch
sys.path.insert(0, '/root/repo')
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
for n in (256, 1024, 2048, 4096, 8192, 16384):
    env = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(0)
    pool = [torch.randn(n, 12, generator=g, device='cuda') * 50 for _ in range(16)]
    for i in range(200): env.step(pool[i % 16])
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for i in range(500): env.step(pool[i % 16])
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 500
    print(f'n={n:6d}  {ms*1e3:7.1f} us/step  {n/ms/1e3:7.2f} M env-steps/s', flush=True)
