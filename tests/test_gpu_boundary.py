"""The MuJoCo entry points of the reference's lower boundary as first-class C-ABI exports: gq_jac (mj_jac,
quadruped_env.py:728), gq_ray (mj_ray, sensors/heightmap.py:90-99), gq_forward (mj_step1 :376 / mj_forward :1321)."""
import numpy as np
import pytest
import torch

from helpers import marshalled, random_states

pytestmark = pytest.mark.gpu


def _env(robot='aliengo', n=32, **kw):
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    kw.setdefault('state_obs_names', ('qpos', 'qvel'))
    return QuadrupedEnv(robot, num_envs=n, device='cuda:0', solver='newton', **kw)


def test_gq_jac_matches_oracle_mj_jac():
    from oracle.oracle import Oracle
    n = 48
    env = _env('aliengo', n)
    md = env.mjModel
    rng = np.random.default_rng(0)
    qpos, qvel = random_states(md, n, rng)
    qpos[:, 0:2] += rng.uniform(-2000, 2000, (n, 2))          # far from the origin: base x/y travel in f64
    o = Oracle(marshalled('aliengo', solver=1))
    for body in (1, 2, 4, 9, 13):
        pts = np.zeros((n, 3))
        for e in range(n):
            o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18)); o.forward(np.zeros(12), stage=1)
            pts[e] = o.xpos[body] + rng.uniform(-0.2, 0.2, 3)
        jp, jr = env.mj_jac(pts, body, qpos=qpos)
        torch.cuda.synchronize()
        for e in range(n):
            o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18)); o.forward(np.zeros(12), stage=1)
            rp, rr = o.jac(pts[e], body)
            np.testing.assert_allclose(jp[e].cpu().numpy(), rp, atol=3e-6)
            np.testing.assert_allclose(jr[e].cpu().numpy(), rr, atol=3e-6)
    # the feet getter of the reference and mj_jac agree (feet_jacobians is mj_jac at the foot geom centre on the calf)
    env.reset(qpos=qpos, qvel=qvel.astype(np.float32))
    with pytest.raises(Exception):
        env.mj_jac(pts, 99)


@pytest.mark.parametrize('scene', ['random_boxes', 'perlin', 'flat'])
def test_gq_ray_matches_numpy_ray_caster(scene):
    """General rays (any origin / direction) against floor, world boxes and the height field vs a brute-force numpy caster."""
    from scipy.spatial.transform import Rotation
    n, R = 16, 24
    env = _env('aliengo', n, scene=scene, seed=3)
    rng = np.random.default_rng(1)
    lim = env.terrain_limits
    org = np.zeros((n, R, 3)); dirs = np.zeros((n, R, 3), np.float32)
    org[..., 0] = rng.uniform(max(lim[1], -6), min(lim[0], 6), (n, R)); org[..., 1] = rng.uniform(max(lim[3], -6), min(lim[2], 6), (n, R))
    org[..., 2] = rng.uniform(0.8, 2.5, (n, R))
    dirs[...] = rng.normal(0, 1, (n, R, 3)); dirs[..., 2] = -np.abs(dirs[..., 2]) - 0.3
    dirs[:, :4] = (0, 0, -1)                                  # the HeightMap's rays
    dirs[:, 4] *= 3.0                                         # not unit length: the distance is in units of |vec|
    dist, geom = env.mj_ray(org, dirs, return_geom=True)
    torch.cuda.synchronize()
    dist, geom = dist.cpu().numpy(), geom.cpu().numpy()
    boxes = env.scene_desc.get('boxes') or []
    Rb = [Rotation.from_quat(np.asarray(b['quat']), scalar_first=True).as_matrix() for b in boxes]
    hf = env.scene_desc.get('hfield')
    tris = None
    if hf is not None:
        data = np.asarray(hf['data'], np.float64) * hf['size'][2]; sx, sy = hf['size'][0], hf['size'][1]; pz = hf.get('pos', (0, 0, 0))[2]
        nr, nc = data.shape
        xs, ys = np.linspace(-sx, sx, nc), np.linspace(-sy, sy, nr)
        P = np.stack([np.tile(xs, (nr, 1)), np.tile(ys[:, None], (1, nc)), data + pz], -1)
        A, B, C, D = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
        tris = np.concatenate([np.stack([A, B, C], -2).reshape(-1, 3, 3), np.stack([D, C, B], -2).reshape(-1, 3, 3)])
    nhit = {0: 0, 1: 0, 2: 0}
    for e in range(0, n, 3):
        for r in range(R):
            o_, d_ = org[e, r], dirs[e, r].astype(np.float64)
            best, kind = (-o_[2] / d_[2], 0) if (d_[2] < 0 and o_[2] >= 0) else (-1.0, -1)
            for b, Rm in zip(boxes, Rb):
                ol, dl = Rm.T @ (o_ - np.asarray(b['pos'])), Rm.T @ d_
                tin, tout, ok = -1e300, 1e300, True
                for k in range(3):
                    if abs(dl[k]) < 1e-14:
                        ok &= abs(ol[k]) <= b['size'][k]; continue
                    t0, t1 = (-b['size'][k] - ol[k]) / dl[k], (b['size'][k] - ol[k]) / dl[k]
                    tin, tout = max(tin, min(t0, t1)), min(tout, max(t0, t1))
                if ok and tin <= tout and tout >= 0:
                    t = tin if tin >= 0 else tout
                    if best < 0 or t < best: best, kind = t, 1
            if tris is not None:   # Moeller-Trumbore on every triangle
                e1, e2 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
                pv = np.cross(d_, e2); det = (e1 * pv).sum(1)
                with np.errstate(divide='ignore', invalid='ignore'):
                    inv = 1.0 / det; tv = o_ - tris[:, 0]; u = (tv * pv).sum(1) * inv
                    qv = np.cross(tv, e1); v = (qv @ d_) * inv; t = (e2 * qv).sum(1) * inv
                ok = (np.abs(det) > 1e-14) & (u >= -1e-9) & (v >= -1e-9) & (u + v <= 1 + 1e-9) & (t >= 0)
                if ok.any() and (best < 0 or t[ok].min() < best): best, kind = t[ok].min(), 2
            assert abs(dist[e, r] - best) < 2e-5 * max(1.0, abs(best)), (scene, e, r, dist[e, r], best)
            assert (geom[e, r] == 0) == (kind == 0) and (geom[e, r] < 0) == (kind < 0)
            nhit[max(kind, 0)] += 1
    if scene == 'random_boxes': assert nhit[1] > 0
    if scene == 'perlin': assert nhit[2] > 20
    # the HeightMap sensor and gq_ray agree on the vertical rays
    from gym_quadruped_amd.sensors import HeightMap
    hm = HeightMap(num_rows=3, num_cols=3, dist_x=0.2, dist_y=0.2, mj_model=env.mjModel, mj_data=env)
    c = torch.as_tensor(org[:, 0], device='cuda:0'); c[:, 2] = 1.0
    pts = hm.update_height_map(c, yaw=0.0).reshape(n, 9, 3)
    o2 = pts.double().clone(); o2[..., 2] = 3.0
    d2 = torch.zeros(n, 9, 3, device='cuda:0'); d2[..., 2] = -1
    z = 3.0 - env.mj_ray(o2, d2)
    assert float((z - pts[..., 2]).abs().max()) < 2e-4


def test_gq_forward_is_mj_forward_without_side_effects():
    """mj_step1 / mj_forward for the batch: nothing but qacc and the inspection record changes; the record holds what the
    reference reads from mjData afterwards (M, qfrc_bias, body poses, contacts, constraint rows, qacc) - vs the oracle."""
    from oracle.oracle import Oracle
    n = 40
    env = _env('aliengo', n, seed=5)
    env.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(0)
    for _ in range(40):
        env.step(torch.randn(n, 12, generator=g, device='cuda:0') * 15)
    torch.cuda.synchronize()
    keep = {k: getattr(env, k).clone() for k in ('_qpos', '_qvel', '_warm', '_time', '_step_num', '_obs_buf', '_terminated')}
    q0, v0, w0, fr = env.qpos.cpu().numpy().copy(), env.qvel.cpu().numpy().copy(), env._warm.cpu().numpy().copy(), env._friction.cpu().numpy().copy()
    ctrl = (np.random.default_rng(2).normal(0, 1, (n, 12)) * 20).astype(np.float32)
    o = Oracle(marshalled('aliengo', solver=1, iterations=100, tolerance=1e-12))
    for stage in (1, 0):
        env.enable_debug(n)
        env.mj_forward(ctrl, stage=stage)
        torch.cuda.synchronize()
        for k, v in keep.items():
            assert torch.equal(getattr(env, k), v), (stage, k)
        dbg = env.debug_internals(n, ['M', 'qfrc_bias', 'nefc', 'ncon', 'efc_J', 'efc_aref', 'efc_R', 'qacc', 'xpos'])
        for e in range(n):
            o.set_state(q0[e], v0[e], w0[e], np.zeros(18), 0.0, float(fr[e])); o.forward(ctrl[e].astype(np.float64), stage=stage)
            d = dbg[e]
            ne = o.nefc
            assert int(d['nefc'][0]) == ne and int(d['ncon'][0]) == o.ncon
            np.testing.assert_allclose(d['M'].reshape(18, 18), o.M, rtol=1e-4, atol=2e-5)
            np.testing.assert_allclose(d['qfrc_bias'], o.qfrc_bias, rtol=1e-4, atol=2e-3)
            np.testing.assert_allclose(d['efc_J'].reshape(64, 18)[:ne], o.efc_J, rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(d['efc_aref'][:ne], o.efc_aref, rtol=2e-4, atol=2e-2)
            xp = d['xpos'].reshape(13, 3); ref = o.xpos[1:] - np.r_[q0[e, :2], 0.0]     # the record is relative to the base x/y
            np.testing.assert_allclose(xp, ref, atol=5e-6)
            if stage == 0:
                assert np.abs(d['qacc'] - o.qacc).max() < 2e-5 * max(1.0, np.abs(o.qacc).max())
                assert np.abs(env._qacc[e].cpu().numpy() - o.qacc).max() < 2e-5 * max(1.0, np.abs(o.qacc).max())
    # the dynamics accessors read the same record: legs_mass_matrix after mj_forward == mj_fullM of the current pose
    Ml = env.legs_mass_matrix
    o.set_state(q0[3], v0[3], w0[3], np.zeros(18), 0.0, float(fr[3])); o.forward(ctrl[3].astype(np.float64), stage=0)
    idx = env.legs_qvel_idx['FL']
    np.testing.assert_allclose(Ml['FL'][3].cpu().numpy(), o.M[np.ix_(idx, idx)], rtol=1e-4, atol=1e-6)
    # ctrl = NULL is zero control (gq_step's contract too)
    env.mj_forward(None, stage=0)
    env.step(torch.zeros(n, 12, device='cuda:0'))
    from gym_quadruped_amd import _lib
    import ctypes as C
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(env._L.gq_step(env._hbatch, None, None, env._st, env._out, None, None, None, stream), 'gq_step with NULL ctrl')
    torch.cuda.synchronize()
    assert torch.isfinite(env.qpos).all()


def test_mj_forward_on_an_accessors_env_refreshes_the_dynamics_rows():
    """An env built with accessors=True answers the dynamics getters from the production kernel's rows and has no inspection record
    until somebody asks for one: mj_forward / mj_step1 must switch the record on themselves (gq_forward writes through it), and the
    getters afterwards describe the FORWARD pose - here a pose that no step has seen."""
    from oracle.oracle import Oracle
    n = 16
    env = _env('aliengo', n, seed=9, accessors=True)
    env.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(3)
    for _ in range(20):
        env.step(torch.randn(n, 12, generator=g, device='cuda:0') * 15)
    M_step = {leg: env.legs_mass_matrix[leg].clone() for leg in ('FL', 'RR')}
    rng = np.random.default_rng(4)
    qpos, qvel = random_states(env.mjModel, n, rng, z_range=(0.35, 0.6))
    env._qpos.copy_(torch.as_tensor(qpos)); env._qvel.copy_(torch.as_tensor(qvel.astype(np.float32)))
    o = Oracle(marshalled('aliengo', solver=1))
    for call in (lambda: env.mj_forward(torch.zeros(n, 12, device='cuda:0')), env.mj_step1):
        call()
        torch.cuda.synchronize()
        Ml, hips = env.legs_mass_matrix, env.hip_positions('world')
        assert not torch.equal(Ml['FL'], M_step['FL'])
        for e in range(n):
            o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18), 0.0, -1.0); o.forward(np.zeros(12), stage=1)
            Mo = np.asarray(o.M).reshape(18, 18)
            for leg in ('FL', 'RR'):
                idx = env.legs_qvel_idx[leg]
                np.testing.assert_allclose(Ml[leg][e].cpu().numpy(), Mo[np.ix_(idx, idx)], rtol=1e-4, atol=1e-6)
    assert torch.equal(env._qpos.cpu(), torch.as_tensor(qpos)), 'a forward pass does not advance the state'


def test_gq_full_mass_is_mj_fullM_of_the_last_forward_pass():
    """gq_full_mass (mj_fullM(model, M, data.qM), quadruped_env.py:557 / :884): the dense joint-space inertia of the last
    forward pass against the oracle's M; symmetric, positive definite; refused when the inspection record is not enabled."""
    import ctypes as C
    from gym_quadruped_amd import _lib
    from oracle.oracle import Oracle
    n = 48
    env = _env('aliengo', n)
    rng = np.random.default_rng(11)
    qpos, qvel = random_states(env.mjModel, n, rng, z_range=(0.3, 0.6))
    env._qpos.copy_(torch.as_tensor(qpos)); env._qvel.copy_(torch.as_tensor(qvel.astype(np.float32)))
    L = _lib.lib()
    M = torch.zeros(n, 18, 18, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    assert L.gq_full_mass(env._hbatch, n, M.data_ptr(), stream) != 0          # no inspection record yet
    env.enable_debug(n)
    env.mj_forward(torch.zeros(n, 12, device='cuda'))
    _lib.check(L.gq_full_mass(env._hbatch, n, M.data_ptr(), stream), 'gq_full_mass')
    torch.cuda.synchronize()
    Mg = M.cpu().numpy()
    o = Oracle(marshalled('aliengo', solver=1))
    for e in range(n):
        o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18), 0.0, -1.0)
        o.forward(np.zeros(12), stage=1)
        Mo = np.asarray(o.M).reshape(18, 18)
        assert np.abs(Mg[e] - Mo).max() < 1e-4 * np.abs(Mo).max(), e
        assert np.abs(Mg[e] - Mg[e].T).max() < 1e-6 * np.abs(Mo).max() and np.linalg.eigvalsh(Mg[e].astype(np.float64)).min() > 0
    assert L.gq_full_mass(env._hbatch, n + 1, M.data_ptr(), stream) != 0      # more envs than the record covers


def test_integration_stub_runs():
    """INTEGRATION.md's binding stub (the ctypes a reference maintainer would write), executed as written against the built
    library on the GPU: create, reset, step, auto-reset step, destroy - and the tensors it leaves are a stepped batch."""
    import os
    from pathlib import Path
    from test_host_and_abi import integration_stub_text
    root = Path(__file__).resolve().parents[1]
    ns = {'__name__': 'integration_stub'}
    cwd = os.getcwd()
    os.chdir(root)
    try:
        exec(compile(integration_stub_text(), 'INTEGRATION.md', 'exec'), ns)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    assert ns['D'] == 73 and bool(torch.isfinite(ns['obs']).all()) and bool(torch.isfinite(ns['qpos']).all())
    assert int(ns['step_num'].min()) >= 1 and float(ns['qpos'][:, 2].min()) > 0.0 and not bool(ns['lift_failed'].any())
    assert float(ns['obs'][:, 37:49].abs().max()) > 1.0   # tau_ctrl_setpoint columns carry the 50 * N(0, 1) torques (clamped to ctrlrange)


def test_bench_process_group_path_initialises_rccl_on_one_rank():
    """bench.py's N > 1 protocol - init_process_group('nccl') = RCCL, barrier + synchronize around the timed region, MAX all-reduce of
    the ranks' times, destroy - executed with ONE rank under torch.distributed.run (the boxes this suite runs on have one GPU; the
    two-rank form of the same protocol runs on gloo in tests/test_sharding_gloo.py): the line it prints is a valid bench line."""
    import json, os, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, GQ_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', '29541',
                        str(root / 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-secondary'],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['scaling'] == 'weak' and d['value'] > 1e7 and d['config']['state_finite']
