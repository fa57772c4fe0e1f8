import sys, time, torch
sys.path.insert(0, '.')
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
n = 4096
g = torch.Generator(device='cuda').manual_seed(0)
pool = [torch.randn(n, 12, generator=g, device='cuda') * 50 for _ in range(16)]
for tag, kw, touch in (('plain', {}, False), ('accessors=True (dyn + contact rows, production kernel)', dict(accessors=True), False),
                       ('inspection record (instrumented kernel)', {}, True)):
    env = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1, **kw)
    env.reset(random=True)
    if touch:
        try: env.legs_mass_matrix
        except Exception: pass
    for i in range(300): env.step(pool[i % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(1000): env.step(pool[i % 16])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{tag}: {n * 1000 / dt / 1e6:.2f} M env-steps/s')
    env.close()
