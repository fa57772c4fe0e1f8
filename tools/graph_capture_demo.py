import sys, torch, numpy as np, time
sys.path.insert(0, '/root/repo')
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
n = 256
def mk(): 
    e = QuadrupedEnv('mini_cheetah', state_obs_names=('qpos', 'qvel'), num_envs=n, auto_reset='next_step', seed=3); e.reset(random=True); return e
a, b = mk(), mk()
act = torch.zeros(n, 12, device='cuda')
g = torch.Generator(device='cuda').manual_seed(0)
seq = [torch.randn(n, 12, generator=g, device='cuda') * 30 for _ in range(40)]
# eager reference
for x in seq: a.step(x)
# graph: one step reading the fixed `act` tensor
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    b.step(act)  # warm (uploads the argument block) - counts as a step with zero action
torch.cuda.synchronize()
b.load_state_dict(mk().state_dict())
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    b.step(act)
b.load_state_dict(mk().state_dict())
for x in seq:
    act.copy_(x); gr.replay()
torch.cuda.synchronize()
print('graph == eager:', torch.equal(a.qpos, b.qpos), torch.equal(a.qvel, b.qvel))
t0 = time.perf_counter()
for i in range(2000): gr.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
for i in range(2000): a.step(act)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'n={n}: graph replay {1e6*(t1-t0)/2000:.1f} us/step, eager {1e6*(t2-t1)/2000:.1f} us/step')
