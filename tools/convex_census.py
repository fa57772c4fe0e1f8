"""How much work the convex narrow phase (oracle/gq_convex.h: GJK + EPA) and the mesh-plane manifold add on benchmark-like states:
N envs of the CPU oracle under 50 N(0,1) N.m random torques with re-spawn on termination (the headline workload's state distribution,
scaled down).  Prints per env-step: convex-routine calls that pass the bounding spheres, contacts, GJK / EPA iterations, the contact
count histogram and the share of env-steps over the kernel's 12-contact capacity.  TEST / MEASUREMENT AID (uses the oracle)."""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from helpers import marshalled, random_states   # noqa: E402
from oracle.oracle import Oracle, lib           # noqa: E402


def main(robot='mini_cheetah', n_envs=48, n_steps=250, boxes=None):
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-8, boxes=boxes)
    md = mm.md
    L = lib()
    L.gqo_cvx_stats.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(0)
    hip = float(mm.desc.key_qpos[2])
    stats = np.zeros(12, dtype=np.int64)
    L.gqo_cvx_stats(None, 1)
    ncon_hist = np.zeros(64, dtype=np.int64)
    call_hist, work_hist, work2_hist = np.zeros(64, dtype=np.int64), np.zeros(256, dtype=np.int64), np.zeros(256, dtype=np.int64)
    prev = np.zeros(12, dtype=np.int64)
    nself_tot = nsteps = nterm = 0
    def spawn():
        q, v = random_states(md, 1, rng, z_range=(1.0 * hip, 1.2 * hip))
        if boxes is not None:   # over the box field (tests/test_kernel_emulated.py test_world_boxes_step_matches_oracle)
            q[0, 0] = rng.uniform(0.5 + 2 * hip, 0.5 + 10 * hip); q[0, 1] = rng.uniform(-3 + 2 * hip, -3 + 12 * hip); q[0, 2] += 0.25 * hip
        return q, v

    for e in range(n_envs):
        o = Oracle(mm)
        q, v = spawn()
        o.set_state(q[0], 0 * v[0], np.zeros(18), np.zeros(18))
        for k in range(n_steps):
            o.step(50.0 * rng.normal(size=12))
            L.gqo_cvx_stats(stats.ctypes.data_as(C.c_void_p), 0)
            d = stats - prev; prev = stats.copy()
            call_hist[min(int(d[0] + d[4]), 63)] += 1
            work_hist[min(int(d[0] + d[4] + d[2] + d[6] + d[3] + d[7]), 255)] += 1
            work2_hist[min(int(d[0] + d[4] + d[2] + d[6] + d[3] + d[7] - d[10]), 255)] += 1   # Minkowski support queries of the env-step: one per call + one per GJK / EPA iteration
            n = int(o.ncon)
            ncon_hist[min(n, 63)] += 1
            g1 = o.get('contact_geom1')[:n]
            nself_tot += int((g1 >= 0).sum())
            nsteps += 1
            bodies = md.geom_bodyid[o.get('contact_geom')[:n].astype(int)]
            world = g1 < 0
            if np.any(world & ((bodies < 2) | ((bodies - 2) % 3 != 2))):   # a non-calf body on the ground: the env terminates and re-spawns
                nterm += 1
                q, v = spawn()
                o.set_state(q[0], 0 * v[0], np.zeros(18), np.zeros(18))
    L.gqo_cvx_stats(stats.ctypes.data_as(C.c_void_p), 1)
    print(f'{robot}: {nsteps} env-steps, {nterm} terminations, self contacts / step {nself_tot / nsteps:.3f}')
    print(f"  self pairs past the bounding spheres / step {stats[8] / nsteps:.2f}")
    for name, s in (("world boxes", stats[:4]), ("self pairs", stats[4:8])):
        calls = max(int(s[0]), 1)
        print(f'  {name}: convex calls / step {s[0] / nsteps:.3f}, contacts / call {s[1] / calls:.3f}, GJK its / call {s[2] / calls:.2f}, EPA its / contact {s[3] / max(int(s[1]), 1):.2f}')
    hist = np.zeros((2, 64), dtype=np.int64)
    L.gqo_cvx_hist.argtypes = [C.c_void_p, C.c_int]
    L.gqo_cvx_hist(hist.ctypes.data_as(C.c_void_p), 1)
    for nm, h in (('GJK iterations per call', hist[0]), ('EPA iterations per contact', hist[1])):
        tot = max(int(h.sum()), 1); cum = np.cumsum(h[::-1])[::-1]
        print(f'  {nm}: ' + ' '.join(f'>={k}:{100.0 * cum[k] / tot:.1f}%' for k in (1, 2, 4, 6, 8, 12, 16, 20, 24, 32) if cum[k]) + f'  max {np.nonzero(h)[0].max() if h.sum() else 0}')
    print(f'  convex contacts that ran into an iteration cap: {stats[9]} of {stats[1] + stats[5]}')
    cs = np.cumsum(call_hist[::-1])[::-1] / call_hist.sum()
    print('  convex calls per env-step: ' + ' '.join(f'>={k}:{100 * cs[k]:.2f}%' for k in (1, 2, 3, 4, 6, 8, 12) if cs[k] > 0))
    ws = np.cumsum(work_hist[::-1])[::-1] / work_hist.sum()
    print('  support queries per env-step: ' + ' '.join(f'>={k}:{100 * ws[k]:.2f}%' for k in (1, 4, 8, 16, 32, 64, 128) if ws[k] > 0), 'max', int(np.nonzero(work_hist)[0].max()))
    ws2 = np.cumsum(work2_hist[::-1])[::-1] / work2_hist.sum()
    print('  ... without the pairs behind a full 12-contact list: ' + ' '.join(f'>={k}:{100 * ws2[k]:.2f}%' for k in (1, 4, 8, 16, 32, 64, 128) if ws2[k] > 0), 'max', int(np.nonzero(work2_hist)[0].max()))
    tot = ncon_hist.sum()
    print('  contacts per env-step: mean %.2f, > 12: %.2f %%, histogram %s' % ((ncon_hist * np.arange(64)).sum() / tot, 100 * ncon_hist[13:].sum() / tot,
                                                                          ' '.join(f'{k}:{100 * c / tot:.1f}' for k, c in enumerate(ncon_hist) if c)))


if __name__ == '__main__':
    # usage: convex_census.py [robot] [boxes]   ('boxes': the random_boxes scene of BASELINE config 5, seed 0)
    robot = sys.argv[1] if len(sys.argv) > 1 else 'mini_cheetah'
    boxes = None
    if len(sys.argv) > 2 and sys.argv[2] == 'boxes':
        from gym_quadruped_amd.terrain import generate_terrain
        from gym_quadruped_amd.robot_cfgs import get_robot_config
        boxes = generate_terrain('random_boxes', float(get_robot_config(robot).hip_height), seed=10)[0]['boxes']
    main(robot, n_envs=24 if boxes is not None else 48, n_steps=150 if boxes is not None else 250, boxes=boxes)
