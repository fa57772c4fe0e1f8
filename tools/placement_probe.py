"""Is the workgroup -> SIMD placement of a 4096-wave launch the same from launch to launch, and does the load of the waves
that share a SIMD decide when the launch ends?   Usage: placement_probe.py [robot] [n]
Instrumented kernel: HW_ID / XCC_ID and the device clock per wave (as tools/wave_timeline.py)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

robot = sys.argv[1] if len(sys.argv) > 1 else 'mini_cheetah'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = QuadrupedEnv(robot, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
for i in range(300): env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
env.enable_debug(n)
slots, ends, nits = [], [], []
for k in range(6):
    env.step(torch.randn(n, 12, generator=g, device='cuda') * 50); torch.cuda.synchronize()
    d = env.debug_internals(n, ['timer', 'niter'])
    T = np.stack([x['timer'] for x in d]); nit = np.array([x['niter'][0] for x in d]).astype(int)
    hw = T[:, 26].astype(int); xcc = T[:, 27].astype(int)
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    slot = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
    t0 = T[:, 24]; t1 = T[:, 25]; base = t0.min()
    ends.append((t1 - base) % (1 << 20) / 100.0); slots.append(slot); nits.append(nit)
for k in range(1, 6):
    print(f'launch {k} vs 0: {np.mean(slots[k] == slots[0]) * 100:.1f} % of the workgroups on the same SIMD; same XCD {np.mean((slots[k] // 512) == (slots[0] // 512)) * 100:.1f} %, same CU {np.mean((slots[k] // 4) == (slots[0] // 4)) * 100:.1f} %')
slot, end, nit = slots[-1], ends[-1], nits[-1]
u, inv = np.unique(slot, return_inverse=True)
load = np.bincount(inv, weights=nit); heavy = np.bincount(inv, weights=(nit >= 3)); fin = np.zeros(len(u)); np.maximum.at(fin, inv, end)
print(f'{robot}: {len(u)} SIMDs; launch ends at {end.max():.1f} us; SIMD finish p50 {np.median(fin):.1f} p90 {np.percentile(fin, 90):.1f} p99 {np.percentile(fin, 99):.1f}')
print('corr(SIMD finish, sum of niter on the SIMD) = %.3f, corr(SIMD finish, max niter on the SIMD) = %.3f' % (np.corrcoef(fin, load)[0, 1], np.corrcoef(fin, np.maximum.reduceat(nit[np.argsort(inv)], np.searchsorted(np.sort(inv), np.arange(len(u)))))[0, 1]))
for h in range(0, 5):
    sel = heavy == h
    if sel.sum(): print(f'  SIMDs with {h} waves of >= 3 iterations: {sel.sum():4d}, finish mean {fin[sel].mean():6.1f} max {fin[sel].max():6.1f} us')
mx = np.zeros(len(u)); np.maximum.at(mx, inv, nit)
for m in range(1, int(mx.max()) + 1):
    sel = mx == m
    if sel.sum():
        print(f'  SIMDs whose slowest wave has {m} iterations: {sel.sum():4d}; finish by #waves with that count: ' + ', '.join(f'{c}: {fin[sel & (np.bincount(inv, weights=(nit == m)) == c)].mean():.1f}' for c in range(1, 5) if (sel & (np.bincount(inv, weights=(nit == m)) == c)).sum()))
print('first 32 workgroups -> (xcd, se, sh, cu, simd):', [(int(s // 512), int(s // 128 % 4), int(s // 64 % 2), int(s // 4 % 16), int(s % 4)) for s in slots[-1][:32]])
