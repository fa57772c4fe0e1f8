import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
for robot in sys.argv[1:]:
    n = 4096
    env = QuadrupedEnv(robot, state_obs_names=('qpos',), num_envs=n, auto_reset='next_step', seed=1000)
    env.reset(random=True)
    g = torch.Generator(device='cuda').manual_seed(0)
    for i in range(200): env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
    env.enable_debug(n)
    env.step(torch.randn(n, 12, generator=g, device='cuda') * 50); torch.cuda.synchronize()
    ptr_t = env._record if hasattr(env, '_record') else None
    env._rec_tensor = None
    rec_n = env._record('niter') if False else None
    import ctypes as C
    from gym_quadruped_amd import _lib
    ptr, nn, st = C.c_void_p(), C.c_int32(), C.c_int32()
    _lib.check(env._L.gq_debug_device_buffer(env._hbatch, C.byref(ptr), C.byref(nn), C.byref(st)), 'buf')
    from gym_quadruped_amd.accessors import _DevPtr
    T = torch.as_tensor(_DevPtr(ptr.value, (nn.value, st.value)), device='cuda')
    def field(name):
        off, cnt = C.c_int32(), C.c_int32(); env._L.gq_debug_field(name.encode(), C.byref(off), C.byref(cnt)); return T[:, off.value:off.value+cnt.value]
    nit = field('niter')[:, 0].cpu().numpy(); nefc = field('nefc')[:, 0].cpu().numpy(); ex = field('timer')[:, 23].cpu().numpy()
    print(robot, 'niter hist', np.bincount(np.minimum(nit.astype(int), 30)).tolist(), 'max', nit.max(), 'nefc mean', nefc.mean(), 'max', nefc.max())
    print('  exit codes', np.bincount(ex.astype(int)).tolist(), ' niter>=20 exits', np.bincount(ex[nit >= 20].astype(int)).tolist() if (nit >= 20).any() else [])
