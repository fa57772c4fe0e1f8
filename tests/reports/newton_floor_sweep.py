"""Sweep of GqModelDesc.noise_floor (fp32 stopping rule of the Newton solver) on rollout states.

States come from CPU oracle rollouts under the bench's action distribution (50*N(0,1) torques), so warm starts are the
previous step's qacc like in the benchmark.  For each floor value the kernel body (host SIMT emulator) takes one step
from every state; reported: mean / max Newton iterations and the qacc error against the CONVERGED fp64 Newton oracle.

    python tests/reports/newton_floor_sweep.py [n_states] [robot]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from helpers import dbg, emu_step, marshalled  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def rollout_states(n, seed=0, robot='mini_cheetah'):
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-10)
    o = Oracle(mm)
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        q = mm.md.key_qpos[0].copy()
        q[7:] += rng.uniform(-0.35, 0.35, 12); q[2] = rng.uniform(0.85, 1.35) * mm.md.key_qpos[0][2]
        o.set_state(q, np.r_[np.zeros(6), rng.uniform(-0.5, 0.5, 12)], np.zeros(18), np.zeros(18), 0.0, -1.0)
        for t in range(int(rng.integers(5, 400))):
            o.step(rng.normal(0, 1, 12).astype(np.float32).astype(np.float64) * 50)
        if not np.all(np.isfinite(o.qpos)):
            continue
        out.append((o.qpos.copy(), o.qvel.copy(), o.qacc_warmstart.copy()))
    return mm, out


def main(n=192, robot='mini_cheetah'):
    _, S = rollout_states(n, robot=robot)
    rng = np.random.default_rng(1)
    qpos = np.stack([s[0] for s in S]); qvel = np.stack([s[1] for s in S]).astype(np.float32); warm = np.stack([s[2] for s in S]).astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 50).astype(np.float32)
    mmN = marshalled(robot, solver=1, iterations=100, tolerance=1e-12)
    oN = Oracle(mmN)
    ref, nit_ref, ncon = [], [], []
    for e in range(n):
        oN.set_state(qpos[e], qvel[e].astype(np.float64), warm[e].astype(np.float64), np.zeros(18), 0.0, -1.0)
        oN.step(ctrl[e].astype(np.float64))
        ref.append(oN.qacc.copy()); nit_ref.append(oN.solver_niter); ncon.append(oN.ncon)
    ref = np.stack(ref); amax = np.maximum(1.0, np.abs(ref).max(1))
    print(f'{n} rollout states, ncon mean {np.mean(ncon):.1f}; fp64 oracle Newton (tol 1e-12): niter mean {np.mean(nit_ref):.2f} max {max(nit_ref)}')
    print('noise_floor   niter mean  p95  max    |qacc-ref|/max|qacc|  p50       p99       max')
    for fl in (0.0, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1e-3):
        mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-8, noise_floor=fl)
        st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), warm=warm.copy(), debug_envs=n)
        qa = np.stack([dbg(st['debug'][e], 'qacc') for e in range(n)])
        ni = np.array([dbg(st['debug'][e], 'niter')[0] for e in range(n)])
        err = np.abs(qa - ref).max(1) / amax
        print(f'{fl:10.0e}   {ni.mean():9.2f} {np.percentile(ni, 95):4.0f} {ni.max():4.0f}    {"":22s}{np.percentile(err, 50):.2e}  {np.percentile(err, 99):.2e}  {err.max():.2e}')


if __name__ == '__main__':
    main(*(int(x) for x in sys.argv[1:2]), *(sys.argv[2:3]))
