"""CPU experiment (oracle, fp64): Newton iteration counts on benchmark-like rollouts for different starting points
(MuJoCo's rule = cheaper of warm start / qacc_smooth; qacc_smooth; zero; warm start).  Usage: python tools/newton_start_experiment.py [robot] [nsteps]
Run as separate processes per mode (the knob is read once per process)."""
import os, subprocess, sys
import numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))

def run(robot, nsteps):
    from helpers import marshalled
    from oracle.oracle import Oracle
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-8)
    o = Oracle(mm)
    q = mm.md.key_qpos[0].copy(); q[2] += 0.05
    o.set_state(q, np.zeros(18), np.zeros(18), np.zeros(18), 0.0, -1.0)
    rng = np.random.default_rng(0)
    hist = np.zeros(40, int); nefc = []
    q0 = q.copy()
    for s in range(nsteps):
        ctrl = rng.normal(0, 1, 12).astype(np.float32).astype(np.float64) * 50
        o.step(ctrl)
        hist[min(o.solver_niter, 39)] += 1; nefc.append(o.nefc)
        _, term, _ = o.get_obs(['qpos'])
        if term:
            o.set_state(q0, np.zeros(18), np.zeros(18), np.zeros(18), 0.0, -1.0)
    k = np.nonzero(hist)[0].max()
    print(f'mode {os.environ.get("GQO_NEWTON_START", "0")} {robot}: mean niter {np.dot(hist, np.arange(40)) / hist.sum():.3f} hist {hist[:k + 1].tolist()} mean nefc {np.mean(nefc):.1f}')

if __name__ == '__main__':
    robot = sys.argv[1] if len(sys.argv) > 1 else 'mini_cheetah'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    if os.environ.get('GQO_CHILD'):
        run(robot, n)
    else:
        for mode in '0123':
            subprocess.run([sys.executable, __file__, robot, str(n)], env=dict(os.environ, GQO_NEWTON_START=mode, GQO_CHILD='1'))
