"""Where and when each wave of one step launch runs (debug kernel: it records a device-wide 100 MHz clock at wave start
and end, the per-stage cycle stamps and the hardware slot).  Shows the dispatch ramp, the end-time distribution, how
waves sharing a SIMD affect each other and which waves form the tail.   Usage: wave_timeline.py [n] [robot]"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
robot = sys.argv[2] if len(sys.argv) > 2 else 'mini_cheetah'
selfcol = None if (len(sys.argv) <= 3 or sys.argv[3] not in ('noself', 'capsule')) else (False if sys.argv[3] == 'noself' else 'capsule')
scene = sys.argv[4] if len(sys.argv) > 4 else 'flat'
env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1, self_collision=selfcol)
env.reset(random=True)
g = torch.Generator(device='cuda').manual_seed(0)
for i in range(300): env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
env.enable_debug(n)
pend = env._terminated.clone().cpu().numpy().astype(bool) if hasattr(env, '_terminated') else np.zeros(n, bool)
env.step(torch.randn(n, 12, generator=g, device='cuda') * 50); torch.cuda.synchronize()
d = env.debug_internals(n, ['timer', 'niter', 'nefc', 'xq'])
T = np.stack([x['timer'] for x in d]); nit = np.array([x['niter'][0] for x in d]).astype(int)
t0 = T[:, 24]; t1 = T[:, 25]
base = t0.min()
t0 = (t0 - base) % (1 << 20) / 100.0; t1 = (t1 - base) % (1 << 20) / 100.0      # us
hint = T[:, 28].astype(int)
hw = T[:, 26].astype(int); xcc = T[:, 27].astype(int)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
slot = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
print(f'{robot}, {n} waves: start min {t0.min():.1f} p50 {np.median(t0):.1f} p99 {np.percentile(t0, 99):.1f} max {t0.max():.1f} us;  end p50 {np.median(t1):.1f} p90 {np.percentile(t1, 90):.1f} p99 {np.percentile(t1, 99):.1f} max {t1.max():.1f} us')
print(f'pending (reset first) waves: {int(pend.sum())};  distinct SIMD slots used: {len(np.unique(slot))}, waves per slot max {np.bincount(np.unique(slot, return_inverse=True)[1]).max()}')
dur = t1 - t0
for k in range(0, nit.max() + 1):
    sel = (nit == k) & ~pend
    if sel.sum(): print(f'  niter {k}: {sel.sum():5d} waves, lifetime mean {dur[sel].mean():6.1f} max {dur[sel].max():6.1f} us, end max {t1[sel].max():6.1f}')
if pend.sum(): print(f'  pending:  {pend.sum():5d} waves, lifetime mean {dur[pend].mean():6.1f} max {dur[pend].max():6.1f} us, end max {t1[pend].max():6.1f}, niter mean {nit[pend].mean():.1f} max {nit[pend].max()}')
for hv in range(4):
    sel = hint == hv
    if sel.sum(): print(f'  hint {hv}: {sel.sum():5d} waves, niter histogram {np.bincount(nit[sel], minlength=7).tolist()}, lifetime mean {dur[sel].mean():6.1f}')
xl = T[:, 31] >= 100; ns = (T[:, 31] % 100).astype(int)
for nm, sel in (('no robot-robot contact', (ns == 0) & ~pend), ('robot-robot contact, tree step', (ns > 0) & ~xl & ~pend), ('cross-leg contact (Sherman-Morrison / dense step)', xl & ~pend)):
    if sel.sum(): print(f'  {nm:50s}: {sel.sum():5d} waves, niter mean {nit[sel].mean():.2f} max {nit[sel].max()}, lifetime mean {dur[sel].mean():6.1f} p99 {np.percentile(dur[sel], 99):6.1f} max {dur[sel].max():6.1f} us, end max {t1[sel].max():6.1f}')
busy = np.array([((t0 <= t) & (t1 > t)).sum() for t in np.arange(0, t1.max(), 2.0)])
print('resident waves every 2 us:', busy.tolist())
# the convex pair exchange as each wave saw it (debug record 'xq', csrc/gq_model_dev.h GQ_DBG_XQ)
XQ = np.stack([x['xq'] for x in d])
rel = lambda col: ((XQ[:, col] - (T[:, 24] - 0)) % (1 << 20)) / 100.0      # us since the wave's start
own = XQ[:, 1] > 0; hlp = XQ[:, 6] > 0; lin = (XQ[:, 9] > 0)
if own.sum():
    print(f'pair exchange: {int((XQ[:, 0] > 0).sum())} envs with convex pairs past the mid phase ({XQ[:, 0].sum():.0f} pairs, max {XQ[:, 0].max():.0f}), {own.sum()} of them published {XQ[own, 1].sum():.0f} (max {XQ[own, 1].max():.0f}), took back {XQ[own, 5].sum():.0f};')
    print(f'  owners, us since their start: pairs READY p50 {np.median(rel(2)[own]):.1f} p99 {np.percentile(rel(2)[own], 99):.1f}; own pairs done p50 {np.median(rel(3)[own]):.1f} p99 {np.percentile(rel(3)[own], 99):.1f}; all DONE p50 {np.median(rel(4)[own]):.1f} p90 {np.percentile(rel(4)[own], 90):.1f} p99 {np.percentile(rel(4)[own], 99):.1f} max {rel(4)[own].max():.1f}')
if hlp.sum():
    per = XQ[hlp, 8] / np.maximum(XQ[hlp, 6], 1) / 100.0
    print(f'  {lin.sum()} envs lingered at the convex block (end p50 {np.median(rel(9)[lin]):.1f} p99 {np.percentile(rel(9)[lin], 99):.1f} us since start); {hlp.sum()} of them computed {XQ[hlp, 6].sum():.0f} pairs for others (max {XQ[hlp, 6].max():.0f}), first claim p50 {np.median(rel(7)[hlp]):.1f} p99 {np.percentile(rel(7)[hlp], 99):.1f} us; us per pair p50 {np.median(per):.1f} p90 {np.percentile(per, 90):.1f} max {per.max():.1f}')
last = np.argsort(-t1)[:12]
print('last waves to finish (end us | start | niter nefc pending | co-resident waves on the SIMD: their niter):')
for e in last:
    co = np.where(slot == slot[e])[0]
    print(f'  env {e:5d}: {t1[e]:6.1f} | {t0[e]:5.1f} | {nit[e]} {int(d[e]["nefc"][0]):2d} {int(pend[e])} hint {hint[e]} self {int(T[e, 31])} | ' + ' '.join(f'{nit[c]}{"r" if pend[c] else ""}h{hint[c]}@{t1[c]:.0f}' for c in co if c != e))
    # stage ends of this wave in us from its start (stage stamps are shader cycles; scaled by the wave's own wall time / cycle count)
    sc = (t1[e] - t0[e]) / max(T[e, 13], 1.0)
    print(f'             exchange: pairs {XQ[e, 0]:.0f} published {XQ[e, 1]:.0f} READY@{rel(2)[e] if XQ[e, 1] else 0:.0f} own done@{rel(3)[e] if XQ[e, 1] else 0:.0f} all DONE@{rel(4)[e] if XQ[e, 1] else 0:.0f} back {XQ[e, 5]:.0f} | helped {XQ[e, 6]:.0f} first@{rel(7)[e] if XQ[e, 6] else 0:.0f} us spent {XQ[e, 8] / 100:.0f} linger end@{rel(9)[e] if XQ[e, 9] else 0:.0f}')
    print('             stage ends (us from wave start): ' + ' '.join(f'{nm} {T[e, k] * sc:.0f}' for nm, k in (('kin', 5), ('floor', 14), ('collide', 6), ('rows', 7), ('pre', 8), ('solve', 9), ('euler', 10), ('integ', 11), ('obs', 12), ('end', 13))))

# the launch's tail as one number (bench.py roofline.tail replays it from profiles/latest_tail.json, with the kernel-source hash it was taken on):
# the share of the launch during which the median wave has already finished, instrumented variant, one launch of the headline workload
import json, os
import bench
tail = {'tail': float((t1.max() - np.median(t1)) / t1.max()), 'launch_us': float(t1.max()), 'median_wave_end_us': float(np.median(t1)), 'p99_wave_end_us': float(np.percentile(t1, 99)),
        'kernel_src_sha16': bench.kernel_source_hash(), 'workload': f'{robot} {scene}, {n} envs, instrumented step kernel, launch 301 of a random-action rollout',
        'profile': 'profiles/r06_wave_timeline.txt'}
print('tail json:', json.dumps(tail))
if robot == 'mini_cheetah' and scene == 'flat' and selfcol is None and n == 4096:
    os.makedirs(ROOT / 'gpurun_out', exist_ok=True)
    (ROOT / 'gpurun_out' / 'latest_tail.json').write_text(json.dumps(tail))
