"""Kernel-time probe: step kernel duration (HIP events) vs PGS iteration cap and batch size."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv

def run(n, iters, tol, steps=300, auto=True, obs='all', solver='pgs'):
    names = tuple(QuadrupedEnv.ALL_OBS) if obs == 'all' else QuadrupedEnv._DEFAULT_OBS
    env = QuadrupedEnv('mini_cheetah', state_obs_names=names, num_envs=n, auto_reset=auto, solver=solver, solver_iterations=iters, solver_tolerance=tol, seed=1)
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(0)
    pool = [torch.randn(n, 12, generator=g, device='cuda') * 50 for _ in range(16)]
    for i in range(50): env.step(pool[i % 16])
    env.enable_debug(min(n, 256))
    env.step(pool[0]); torch.cuda.synchronize()
    nit = np.mean([d['niter'][0] for d in env.debug_internals(min(n, 256), ['niter'])])
    nefc = np.mean([d['nefc'][0] for d in env.debug_internals(min(n, 256), ['nefc'])])
    env.enable_debug(0)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); s.record()
    nterm = 0
    for i in range(steps):
        o, r, term, tr, info = env.step(pool[i % 16])
    e.record(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    ms = s.elapsed_time(e) / steps
    print(f'{solver} n={n:5d} iters={iters:3d} tol={tol:g} auto={auto} obs={obs}: {ms*1e3:7.1f} us/step (gpu) {wall/steps*1e6:7.1f} us/step (wall)  {n/ms/1e3:6.2f} M env-steps/s  mean niter {nit:.1f} nefc {nefc:.1f} term/step {float(term.float().mean()):.4f}')

if __name__ == '__main__' and len(sys.argv) == 1:
    for it, tol in [(0, 0), (10, 0), (50, 0), (100, 1e-8)]:
        run(4096, it, tol)
    run(4096, 100, 1e-8, auto=False)
    run(4096, 100, 1e-8, solver='newton')
    run(1024, 100, 1e-8, solver='newton')
    run(16384, 100, 1e-8, solver='newton')
    run(4096, 100, 1e-8, obs='default')
    for n in (1024, 2048, 8192, 16384):
        run(n, 100, 1e-8)


def stage_times(n=4096, iters=100, tol=1e-8, solver='newton', robot='mini_cheetah', scene='flat'):
    env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', solver=solver, solver_iterations=iters, solver_tolerance=tol, seed=1)
    env.reset(random=True)
    g = torch.Generator(device='cuda').manual_seed(0)
    for i in range(300): env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)   # the benchmark's steady state
    env.enable_debug(n)
    env.step(torch.randn(n, 12, generator=g, device='cuda') * 50); torch.cuda.synchronize()
    d = env.debug_internals(n, ['timer', 'niter', 'nefc'])
    T = np.stack([x['timer'] for x in d]); nit = np.array([x['niter'][0] for x in d]); ne = np.array([x['nefc'][0] for x in d])
    names = ['S0+S1 kin', 'S2 inert', 'S3 M', 'S4 factor', 'S5 rne', 'S6a scan', 'S6b list', 'S7 rows', 'S8 dual/qs', 'S9 solver', 'S10 acc', 'dump+euler', 'S11 obs', 'gather']
    order = [1, 2, 3, 4, 5, 14, 6, 7, 8, 9, 10, 11, 12, 13]
    prev = np.zeros(n)
    print(f'stage times in shader cycles (mean / p95 / max over {n} envs); niter mean {nit.mean():.1f} p95 {np.percentile(nit,95):.0f} max {nit.max():.0f}; nefc mean {ne.mean():.1f} max {ne.max():.0f}')
    for nm, k in zip(names, order):
        dt = T[:, k] - prev; prev = T[:, k]
        print(f'  {nm:10s} {dt.mean():9.0f} {np.percentile(dt,95):9.0f} {dt.max():9.0f}')
    print(f'  total      {T[:,13].mean():9.0f} {np.percentile(T[:,13],95):9.0f} {T[:,13].max():9.0f}')
    top = np.argsort(-T[:, 13])[:10]
    print('  slowest waves (env: total | per-stage ... | niter nefc):')
    for e in top:
        prev = 0.0; parts = []
        for k in order:
            parts.append(T[e, k] - prev); prev = T[e, k]
        print(f'   env {e:5d}: {T[e,13]:8.0f} | ' + ' '.join(f'{x:6.0f}' for x in parts) + f' | niter {nit[e]:.0f} nefc {ne[e]:.0f} self {T[e,31]:.0f} ls trials {T[e,29]:.0f} full {T[e,30]:.0f} exit {T[e,23]:.0f} | newton parts ' + ' '.join(f'{T[e,j]:.0f}' for j in range(16, 23)))
    h, edges = np.histogram(T[:, 13], bins=12)
    print('  histogram of totals:', ' '.join(f'{int(lo/1000)}k:{c}' for lo, c in zip(edges[:-1], h)))
    hn = np.bincount(nit.astype(int)); print('  niter histogram:', hn.tolist())
    if solver == 'newton':
        for nm, k in zip(['setup', 'state+cost', 'gradient', 'hessian', 'solve', 'ls prep', 'ls trials+upd'], range(16, 23)):
            print(f'    newton {nm:13s} {T[:, k].mean():9.0f} {np.percentile(T[:, k], 95):9.0f} {T[:, k].max():9.0f}   per-iter {T[:, k].sum() / max(1, nit.sum()):7.0f}')
        print(f'    line-search trials per iteration {T[:, 29].sum() / max(1, nit.sum()):.2f}; full-step shortcuts per iteration {T[:, 30].sum() / max(1, nit.sum()):.2f}')
        xl = T[:, 31] >= 100; ns = T[:, 31] % 100
        for nm, sel in (('no robot-robot contact', ns == 0), ('robot-robot contact, tree step', (ns > 0) & ~xl), ('cross-leg contact, dense step', xl)):
            if sel.sum():
                print(f'    {nm:32s}: {sel.sum():5d} waves, niter mean {nit[sel].mean():.2f} max {nit[sel].max():.0f}, nefc mean {ne[sel].mean():.1f}, total {T[sel, 13].mean():8.0f} cycles, '
                      f'S6b {(T[sel, 6] - T[sel, 14]).mean():6.0f} S7 {(T[sel, 7] - T[sel, 6]).mean():6.0f} solver {(T[sel, 9] - T[sel, 8]).mean():7.0f}; per iteration: hessian {T[sel, 19].sum() / nit[sel].sum():6.0f} solve {T[sel, 20].sum() / nit[sel].sum():6.0f}')
        W0 = T[:, 0].astype(int); T = T.copy(); T[:, 24], T[:, 26], T[:, 27] = W0 & 63, (W0 >> 6) & 63, W0 >> 12   # the dense-step census, packed in slot 0 (gq_newton.h)
        dn = T[:, 26] > 0
        if dn.sum():   # dense steps: how many cross-leg rows were active (what a low-rank correction of the tree solve would have to carry)
            kmax = T[dn, 24].astype(int)
            print(f'    dense steps: {int(T[:, 26].sum())} in {dn.sum()} waves (+ {int(T[:, 27].sum())} Sherman-Morrison steps); active cross-leg rows in a dense step, '
                  f'largest per wave: ' + ' '.join(f'{k}:{c}' for k, c in enumerate(np.bincount(kmax)) if c))
            slow = np.argsort(-T[:, 13])[:64]
            print('    among the 64 slowest waves: largest active cross-leg row count per wave ' + ' '.join(f'{k}:{c}' for k, c in enumerate(np.bincount(T[slow, 24].astype(int))) if c) + f'; dense steps {int(T[slow, 26].sum())}, SM steps {int(T[slow, 27].sum())}')
        for k in range(2, int(nit.max()) + 1):
            sel = nit == k
            if sel.sum():
                print(f'    niter {k}: {sel.sum():4d} waves, solver stage {(T[sel, 9] - T[sel, 8]).mean():8.0f} cycles, per part ' + ' '.join(f'{T[sel, j].mean():7.0f}' for j in range(16, 23)) + f' | ls trials {T[sel, 29].mean():.1f} full {T[sel, 30].mean():.1f}')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'stages':
    robot = [x for x in sys.argv[2:] if not x.isdigit()]
    for n in (int(x) for x in ([x for x in sys.argv[2:] if x.isdigit()] or ['4096'])):
        for sv in ('newton',):
            print(sv, robot); stage_times(n, solver=sv, robot=(robot or ['mini_cheetah'])[0], scene=(robot + ['flat', 'flat'])[1])


SUBSETS = {
    3: ['entry -> pointer batch in', 'vector loads issued + model scalars', 'rows in LDS (all loads landed)', 'step_wave entry', 'records issued, barrier', '-', '-', '-', '-', '-', '-', '-', '-', '(rest)'],
    1: ['entry->rows in LDS', 'actuation+passive', 'kin phase 1 (lane=link)', 'kin phase 2 (chains)', 'S2 inertias', 'S3 mass matrix', 'S5 rne',
        'S6a hull scan', 'floor candidates', 'floor list + limits', 'self: end points', 'self: pair cull', 'self: rest', '(S7 .. end)'],
    2: ['(entry .. S6)', 'S7 row descriptors', 'S7 J sweep', 'S7 impedance/aref', 'S9 solver', 'S10 Euler system', 'integrate + state stores',
        'S11 base obs', 'S11 joints + energy', 'S11 feet + forces', 'S11 imu/term/flags', 'gather', 'resample', '(end)'],
}


def substage_times(which, n=256, robot='mini_cheetah', scene='flat'):
    """Sub-stage cut of a library built with -DGQ_TICKSET=<which> (tools/dev_build.sh): the 13 stage stamps sit at the GQ_SUB points of that set."""
    env = QuadrupedEnv(robot, scene=scene, state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, auto_reset='next_step', seed=1)
    env.reset(random=True)
    g = torch.Generator(device='cuda').manual_seed(0)
    for i in range(300): env.step(torch.randn(n, 12, generator=g, device='cuda') * 50)
    env.enable_debug(n)
    env.step(torch.randn(n, 12, generator=g, device='cuda') * 50); torch.cuda.synchronize()
    d = env.debug_internals(n, ['timer', 'niter', 'nefc'])
    T = np.stack([x['timer'] for x in d]); nit = np.array([x['niter'][0] for x in d])
    order = [1, 2, 3, 4, 5, 14, 6, 7, 8, 9, 10, 11, 12, 13]
    print(f'sub-stage set {which}, {robot} {scene}, {n} envs: cycles from kernel entry (mean / p95 / max); niter mean {nit.mean():.2f}')
    prev = np.zeros(n)
    for nm, k in zip(SUBSETS[which], order):
        dt = T[:, k] - prev; prev = T[:, k]
        print(f'  {nm:28s} {dt.mean():9.0f} {np.percentile(dt, 95):9.0f} {dt.max():9.0f}')
    print(f'  total                        {T[:, 13].mean():9.0f}')


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[1] == 'sub':
    substage_times(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 256, *(sys.argv[4:6]))
