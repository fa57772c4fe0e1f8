"""Parity report: kernel (GPU through the C-ABI, or the host SIMT emulator with --emu) vs the CPU oracle on N random
states.  Prints error percentiles of qacc / qvel / qpos against (a) the oracle running the same PGS iteration count
and (b) the oracle's converged Newton solution (the reference's default solver)."""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from helpers import ALL_OBS, dbg, emu_step, marshalled, random_states  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--tol', type=float, default=0.0)
    ap.add_argument('--emu', action='store_true')
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    mm = marshalled('mini_cheetah', solver=0, iterations=a.iters, tolerance=a.tol)
    mmN = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-12)
    rng = np.random.default_rng(a.seed)
    qpos, qvel = random_states(mm.md, a.n, rng)
    qvel = qvel.astype(np.float32)
    warm = rng.normal(0, 5, (a.n, 18)).astype(np.float32)
    ctrl = (rng.normal(0, 1, (a.n, 12)) * 30).astype(np.float32)
    if a.emu:
        st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), warm=warm.copy(), debug_envs=a.n)
        qacc_k = np.stack([dbg(st['debug'][e], 'qacc') for e in range(a.n)])
        niter = np.array([dbg(st['debug'][e], 'niter')[0] for e in range(a.n)])
        qvel_k, qpos_k = st['qvel'], st['qpos']
    else:
        import torch
        from gym_quadruped_amd.quadruped_env import QuadrupedEnv
        env = QuadrupedEnv('mini_cheetah', state_obs_names=('qpos',), num_envs=a.n, solver='pgs', solver_iterations=a.iters, solver_tolerance=a.tol)
        env._qpos.copy_(torch.as_tensor(qpos)); env._qvel.copy_(torch.as_tensor(qvel)); env._warm.copy_(torch.as_tensor(warm))
        env.enable_debug(a.n)
        env.step(torch.as_tensor(ctrl))
        torch.cuda.synchronize()
        d = env.debug_internals(a.n, ['qacc', 'niter'])
        qacc_k = np.stack([x['qacc'] for x in d]); niter = np.array([x['niter'][0] for x in d])
        qvel_k, qpos_k = env.qvel.cpu().numpy(), env.qpos.cpu().numpy()
    o, oN = Oracle(mm), Oracle(mmN)
    e_acc, e_accN, e_vel, e_pos, amax, ncon = [], [], [], [], [], []
    for e in range(a.n):
        o.set_state(qpos[e], qvel[e], warm[e], np.zeros(18), 0.0, -1.0); o.step(ctrl[e].astype(np.float64))
        oN.set_state(qpos[e], qvel[e], warm[e], np.zeros(18), 0.0, -1.0); oN.step(ctrl[e].astype(np.float64))
        e_acc.append(np.abs(qacc_k[e] - o.qacc).max()); e_accN.append(np.abs(qacc_k[e] - oN.qacc).max())
        e_vel.append(np.abs(qvel_k[e] - o.qvel).max()); e_pos.append(np.abs(qpos_k[e] - o.qpos).max())
        amax.append(np.abs(o.qacc).max()); ncon.append(o.ncon)
    e_acc, e_accN, e_vel, e_pos, amax, ncon = map(np.array, (e_acc, e_accN, e_vel, e_pos, amax, ncon))
    pct = lambda x: ' '.join(f'{np.percentile(x, p):.2e}' for p in (50, 90, 99, 100))
    print(f'n={a.n} iters={a.iters} tol={a.tol} kernel={"emulator" if a.emu else "gpu"}; mean niter {niter.mean():.1f}; ncon mean {ncon.mean():.1f}')
    print('percentiles                      p50      p90      p99      max')
    print('|qacc - oracle PGS|/max|qacc|  ', pct(e_acc / np.maximum(1, amax)))
    print('|qacc - oracle Newton|/max|qacc|', pct(e_accN / np.maximum(1, amax)))
    print('|qvel - oracle PGS|            ', pct(e_vel))
    print('|qpos - oracle PGS|            ', pct(e_pos))
    w = int(np.argmax(e_acc / np.maximum(1, amax)))
    print('worst env', w, 'ncon', ncon[w], 'amax', amax[w], 'err', e_acc[w])


if __name__ == '__main__':
    main()
