"""The UNMODIFIED HIP kernel bodies (csrc/gq_step_body.h) executed by the host SIMT emulator (tests/simt_emu) against
the CPU oracle: same parity contract as tests/test_gpu_parity.py, runnable without a GPU."""
from pathlib import Path
import ctypes as C
import numpy as np
import pytest

from helpers import ALL_OBS, ParityTally, dbg, oracle_fits_row_budget, default_reset_cfg, emu_reset, emu_step, marshalled, oracle_fits_self_budget, oracle_reset_lift, random_states, self_contact_states, split_obs
from oracle.oracle import Oracle
from philox_ref import draws


def test_step_stagewise_and_obs():
    n = 24
    mm = marshalled('mini_cheetah', solver=0, iterations=30, tolerance=0.0)
    rng = np.random.default_rng(11)
    qpos, qvel = random_states(mm.md, n, rng)
    qvel = qvel.astype(np.float32)
    warm = rng.normal(0, 5, (n, 18)).astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 40).astype(np.float32)   # beyond ctrlrange: exercises the clamp
    cmd = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    fric = np.where(np.arange(n) % 2 == 0, -1.0, 0.55).astype(np.float32)
    lo = (2, 0, 3, 1)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), warm=warm.copy(), cmd=cmd, friction=fric, legs_order=lo, debug_envs=n)
    o = Oracle(mm)
    ncon_total = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e], warm[e], np.zeros(18), 0.0, float(fric[e]))
        o.step(ctrl[e].astype(np.float64))
        if o.ncon and o.get('contact_tiegap').min() < 3e-7:
            continue
        rec = st['debug'][e]
        ne = o.nefc
        if not oracle_fits_row_budget(o, False):   # (a robot pressed into the floor: its mesh manifolds exceed the kernel's 12 contacts / 63 rows; the budget tests hold those to the prefix rule)
            assert int(dbg(rec, 'nefc')[0]) < ne
            continue
        ncon_total += o.ncon
        assert int(dbg(rec, 'nefc')[0]) == ne and int(dbg(rec, 'ncon')[0]) == o.ncon
        np.testing.assert_allclose(dbg(rec, 'M').reshape(18, 18), o.M, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dbg(rec, 'qfrc_bias'), o.qfrc_bias, rtol=1e-5, atol=5e-4)
        np.testing.assert_allclose(dbg(rec, 'efc_J').reshape(64, 18)[:ne], o.efc_J, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dbg(rec, 'efc_R')[:ne], o.efc_R, rtol=1e-5)
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:ne], o.efc_aref, rtol=1e-4, atol=1e-2)
        fmax = max(1.0, np.abs(o.efc_force).max())
        assert np.abs(dbg(rec, 'efc_force')[:ne] - o.efc_force).max() < 1e-4 * fmax
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-5 * max(1.0, np.abs(o.qacc).max())
        assert np.abs(st['qvel'][e] - o.qvel).max() < 5e-5
        assert np.abs(st['qpos'][e] - o.qpos).max() < 1e-6
        ref, term, inv = o.get_obs(ALL_OBS, cmd[e], lo)
        got = split_obs(st['obs'][e], ALL_OBS)
        for k in ALL_OBS:
            assert np.abs(got[k] - ref[k]).max() < 2e-4 * max(1.0, np.abs(ref[k]).max()), (e, k)
        assert bool(st['terminated'][e]) == term and bool(st['invalid'][e]) == inv
        assert st['step_num'][e] == 1 and abs(st['time'][e] - 0.002) < 1e-9 and st['reward'][e] == 0
    assert ncon_total > 30


def test_mask_leaves_other_envs_untouched():
    mm = marshalled('mini_cheetah', iterations=5)
    rng = np.random.default_rng(2)
    qpos, qvel = random_states(mm.md, 4, rng)
    q0, v0 = qpos.copy(), qvel.astype(np.float32).copy()
    st = emu_step(mm, np.zeros((4, 12)), qpos, qvel, mask=[1, 0, 0, 1])
    assert np.array_equal(st['qpos'][1], q0[1]) and np.array_equal(st['qvel'][2], v0[2])
    assert not np.array_equal(st['qpos'][0], q0[0]) and st['step_num'].tolist() == [1, 0, 0, 1]


def test_out_of_bounds_terminates_far_from_origin_in_f64():
    """Base x/y are carried in f64: 1 mm steps at |x| = 9999.9995 m are resolved and the terrain limit trips."""
    mm = marshalled('mini_cheetah', iterations=5)
    q = np.tile(mm.md.key_qpos[0], (2, 1)); q[:, 2] = 1.0
    q[0, 0] = 9999.9995; q[1, 0] = 9999.0
    v = np.zeros((2, 18), np.float32); v[:, 0] = 0.5
    st = emu_step(mm, np.zeros((2, 12)), q, v)
    assert abs(st['qpos'][0, 0] - (9999.9995 + 0.002 * 0.5)) < 1e-9
    assert st['terminated'].tolist() == [1, 0]


def test_reset_kernel_stream_lift_and_bookkeeping():
    mm = marshalled('mini_cheetah', terrain_limits=(5, -5, 5, -5))
    seed = 987654321987
    cfg = default_reset_cfg(seed=seed, lin=(0.5, 1.0), ang=(-0.3, 0.3), fric=(0.2, 1.5), cmd='random+rotate')
    n = 6
    st = emu_reset(mm, n, cfg, episode=np.arange(n))
    o = Oracle(mm)
    key = mm.md.key_qpos[0]
    for e in range(n):
        u = draws(seed, e, e)
        np.testing.assert_allclose(st['qpos'][e, 7:], key[7:] + (2 * u[0:12] - 1) * np.float32(20 * np.pi / 180), atol=1e-6)
        np.testing.assert_allclose(st['qvel'][e, 6:], (2 * u[12:24] - 1) * 0.5, atol=1e-6)
        assert abs(st['qpos'][e, 0] - (5 - 10 * float(u[24]))) < 1e-9 and abs(st['qpos'][e, 1] - (5 - 10 * float(u[25]))) < 1e-9
        norm, head = 0.5 + 0.5 * u[28], (2 * u[29] - 1) * np.pi
        np.testing.assert_allclose(st['cmd'][e], [norm * np.cos(head), norm * np.sin(head), 0, -0.3 + 0.6 * u[30]], atol=1e-5)
        assert abs(st['friction_next'][e] - (0.2 + 1.3 * u[31])) < 1e-6
        assert st['episode'][e] == e + 1 and st['step_num'][e] == -1 and st['time'][e] == 0
        assert not st['qacc'][e].any() and not st['warm'][e].any() and not st['applied'][e].any()
        # flat scene: the reset kernel writes the spawn pose at hip height and leaves word that the lift loop is due; it
        # runs inside the reset's own mj_step (below)
        assert st['lift_pending'][e] == 1 and abs(st['qpos'][e, 2] - 0.225) < 1e-7
    spawn_q, spawn_v = st['qpos'].copy(), st['qvel'].copy()
    st1 = emu_step(mm, np.zeros((n, 12)), st['qpos'], st['qvel'], warm=st['warm'], applied=st['applied'], time=st['time'],
                   cmd=st['cmd'], step_num=st['step_num'], episode=st['episode'], first_pass=1, lift_pending=st['lift_pending'],
                   friction_next=st['friction_next'], obs_names=['qpos', 'feet_pos'])
    nlift = 0
    for e in range(n):
        # lift loop on the oracle (the reference's rule), then the reset's mj_step from the lifted pose
        z, it = oracle_reset_lift(o, spawn_q[e], spawn_v[e].astype(np.float64), 0.225)
        nlift += it > 0
        q0 = spawn_q[e].copy(); q0[2] = z
        o.set_state(q0, spawn_v[e].astype(np.float64), np.zeros(18), np.zeros(18), 0.0, -1.0); o.step(np.zeros(12))
        assert not st1['lift_failed'][e] and st1['step_num'][e] == 0
        assert np.abs(st1['qpos'][e] - o.qpos).max() < 1e-6 and np.abs(st1['qvel'][e] - o.qvel).max() < 5e-5, (e, it)
        ref, _, _ = o.get_obs(['feet_pos'])
        assert np.abs(split_obs(st1['obs'][e], ['qpos', 'feet_pos'])['feet_pos'] - ref['feet_pos']).max() < 2e-5   # old kinematics AT THE LIFTED POSE
        assert abs(st1['friction'][e] - st['friction_next'][e]) < 1e-9      # committed after the reset's step (:403-404)
    assert nlift > 0
    # explicit state: copied verbatim, no lift
    qn = np.tile(key, (2, 1)); qn[:, 2] = 0.05
    st2 = emu_reset(mm, 2, cfg, qpos_new=qn, qvel_new=np.ones((2, 18), np.float32))
    assert np.array_equal(st2['qpos'], qn) and np.all(st2['qvel'] == 1)


def test_imu_truth_noise_and_bias_walk():
    """IMU observables out of the step kernel: ground truth == oracle's mj_sensorAcc/Vel restatement, noise and bias
    random walk == the documented Philox normal stream (tests/philox_ref.py), draw order of sensors/imu.py:110-139."""
    import ctypes as C
    from helpers import IMU_OBS, GqImuCfg
    from philox_ref import imu_normals
    mm = marshalled('aliengo', solver=1, iterations=100, tolerance=1e-8)
    o = Oracle(marshalled('aliengo', solver=1, iterations=100, tolerance=1e-12))
    pos, quat = (0.05, -0.02, 0.03), np.array([0.9, 0.1, -0.3, 0.2])
    quat = quat / np.linalg.norm(quat)
    o.set_imu(pos, quat)
    imu = GqImuCfg(site_pos=(C.c_double * 3)(*pos), site_quat=(C.c_double * 4)(*quat), accel_noise=0.01, gyro_noise=0.02,
                   accel_bias_rate=0.03, gyro_bias_rate=0.04, seed=777)
    rng = np.random.default_rng(5)
    n = 8
    qpos, qvel = random_states(mm.md, n, rng, z_range=(0.3, 0.5))
    qvel = qvel.astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    b0 = rng.normal(0, 0.1, (n, 6)).astype(np.float32)
    names = ['qpos'] + list(IMU_OBS)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), obs_names=names, imu=imu, imu_bias=b0.copy(), step_num=np.arange(n) + 10,
                  episode=np.arange(n))
    compared = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18), 0, -1)
        o.step(ctrl[e].astype(np.float64))
        if not oracle_fits_self_budget(o, mm.md.cone):
            continue   # more contacts than one wavefront's rows: the accelerations differ by construction
        compared += 1
        got = split_obs(st['obs'][e], names)
        z = imu_normals(777, e, 10 + e, e)
        an, ab, gn, gb = z[0:3] * 0.01, b0[e, :3] + z[3:6] * 0.03, z[6:9] * 0.02, b0[e, 3:] + z[9:12] * 0.04
        tol = 2e-5 * max(1.0, np.abs(o.imu_acc).max())
        assert np.abs(got['imu_acc'] - (o.imu_acc + ab + an)).max() < tol
        assert np.abs(got['imu_gyro'] - (o.imu_gyro + gb + gn)).max() < 1e-5
        np.testing.assert_allclose(got['imu_acc_noise'], an, atol=1e-7); np.testing.assert_allclose(got['imu_gyro_noise'], gn, atol=1e-7)
        np.testing.assert_allclose(got['imu_acc_bias'], ab, atol=1e-6); np.testing.assert_allclose(got['imu_gyro_bias'], gb, atol=1e-6)
        np.testing.assert_allclose(st['imu_bias'][e], np.r_[ab, gb], atol=1e-6)
    assert compared >= n // 2
    # free fall: the accelerometer reads zero (and +g once something holds the base still: a = 0 -> reading = -gravity)
    mm2 = marshalled('aliengo', solver=1)
    o2 = Oracle(mm2)
    q = mm2.md.key_qpos[0].copy(); q[2] = 2.0
    o2.set_state(q, np.zeros(18), np.zeros(18), np.zeros(18))
    o2.forward(np.zeros(12))
    assert np.abs(o2.imu_acc).max() < 1e-9 and np.abs(o2.imu_gyro).max() == 0
    hold = np.zeros(18); hold[2] = mm2.md.total_mass * 9.81          # external force cancelling the weight
    o2.set_state(q, np.zeros(18), np.zeros(18), hold)
    o2.forward(np.zeros(12))
    assert np.abs(o2.imu_acc - [0, 0, 9.81]).max() < 0.5               # legs still sag, the base barely moves


def test_auto_reset_same_step_and_next_step_agree():
    """A terminated env (base out of bounds) is re-spawned either inside the same launch (second pass) or by its next
    launch (gymnasium NEXT_STEP): both run reset_wave + the reset's own mj_step on the same RNG counters, so the state
    after 'step' (same-step) equals the state after 'step, step' (next-step); the untouched env just steps."""
    mm = marshalled('mini_cheetah', iterations=20, terrain_limits=(5, -5, 5, -5))
    q = np.tile(mm.md.key_qpos[0], (2, 1)); q[:, 2] = 0.6
    q[0, 0] = 5.5    # env 0 is outside the terrain -> terminates
    v = np.zeros((2, 18), np.float32)
    ctrl = np.ones((2, 12), np.float32)
    epi = np.array([3, 4], np.int32)
    cfg_same = default_reset_cfg(seed=77, fric=(0.3, 0.9))
    cfg_next = default_reset_cfg(seed=77, fric=(0.3, 0.9)); cfg_next.autoreset_next_step = 1
    a = emu_step(mm, ctrl, q.copy(), v.copy(), auto_reset=cfg_same, episode=epi.copy())
    assert a['terminated'].tolist() == [1, 0] and a['episode'].tolist() == [4, 4] and a['pending'].tolist() == [0, 0]
    assert a['step_num'].tolist() == [0, 1] and abs(a['qpos'][0, 0]) <= 5.0
    b1 = emu_step(mm, ctrl, q.copy(), v.copy(), auto_reset=cfg_next, episode=epi.copy())
    assert b1['terminated'].tolist() == [1, 0] and b1['pending'].tolist() == [1, 0] and b1['episode'].tolist() == [3, 4]
    assert abs(b1['qpos'][0, 0] - 5.5) < 1e-4 and b1['step_num'].tolist() == [1, 1]      # terminal state / observation kept
    np.testing.assert_array_equal(b1['qpos'][1], a['qpos'][1])
    q1_prev = b1['qpos'][1].copy()   # emu_step updates the arrays in place
    b2 = emu_step(mm, ctrl, b1['qpos'], b1['qvel'], warm=b1['warm'], time=b1['time'], friction=b1['friction'], cmd=b1['cmd'],
                  auto_reset=cfg_next, episode=b1['episode'], pending=b1['pending'], step_num=b1['step_num'])
    assert b2['terminated'].tolist() == [0, 0] and b2['pending'].tolist() == [0, 0] and b2['episode'].tolist() == [4, 4]
    assert b2['step_num'].tolist() == [0, 2]
    for k in ('qpos', 'qvel', 'warm', 'time', 'friction', 'cmd'):
        np.testing.assert_array_equal(b2[k][0], a[k][0], err_msg=k)   # same draws, same reset step: bit-identical
    np.testing.assert_array_equal(b2['obs'][0], a['obs'][0])
    assert not np.array_equal(b2['qpos'][1], q1_prev)          # env 1 kept stepping with the user's control


@pytest.mark.parametrize('robot', ['hyqreal1', 'go2'])
def test_elliptic_cone_step_matches_converged_oracle(robot):
    """Elliptic friction cones (cone="elliptic", impratio 100; go2 feet are condim 6: torsional + rolling rows): rows,
    regularisation and the Newton solution of the kernel against the fp64 oracle converged to 1e-12."""
    n = 24
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-8, noise_floor=1e-5)
    mmN = marshalled(robot, solver=1, iterations=200, tolerance=1e-13)
    rng = np.random.default_rng(5)
    hip = float(mm.desc.key_qpos[2])
    qpos, qvel = random_states(mm.md, n, rng, z_range=(0.9 * hip, 1.2 * hip))   # feet and a few links down; a robot lying
    qvel = qvel.astype(np.float32)                                               # flat exceeds the 64-row budget (and is terminated)
    warm = rng.normal(0, 3, (n, 18)).astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    fric = np.where(np.arange(n) % 2 == 0, -1.0, 0.6).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), warm=warm.copy(), friction=fric.copy(), debug_envs=n)
    o = Oracle(mmN)
    ncon = nchecked = 0
    tally = ParityTally(cone=True, tie_threshold=3e-7)
    for e in range(n):
        o.set_state(qpos[e], qvel[e].astype(np.float64), warm[e].astype(np.float64), np.zeros(18), 0.0, float(fric[e]))
        o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        nefc = int(dbg(rec, 'nefc')[0])
        if tally.classify(e, o, nefc) != 'ok':
            continue
        nchecked += 1; ncon += o.ncon
        J = dbg(rec, 'efc_J').reshape(64, 18)[:nefc]
        np.testing.assert_allclose(J, o.efc_J, atol=2e-5 * max(1.0, np.abs(o.efc_J).max()))
        np.testing.assert_allclose(dbg(rec, 'efc_R')[:nefc], o.efc_R, rtol=2e-4)
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:nefc], o.efc_aref, atol=2e-4 * max(1.0, np.abs(o.efc_aref).max()))
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-4 * max(1.0, np.abs(o.qacc).max()), (e, dbg(rec, 'niter'))
        fmax = max(1.0, np.abs(o.efc_force).max())
        assert np.abs(dbg(rec, 'efc_force')[:nefc] - o.efc_force).max() < 2e-3 * fmax
        assert np.abs(st['qvel'][e] - o.qvel).max() < 5e-4 and np.abs(st['qpos'][e] - o.qpos).max() < 2e-6
        ref, t, inv = o.get_obs(ALL_OBS, np.zeros(4))
        got = split_obs(st['obs'][e], ALL_OBS)
        for k in ('contact_forces:base', 'contact_forces', 'feet_vel', 'contact_state'):
            assert np.abs(got[k] - ref[k]).max() < 2e-3 * max(1.0, np.abs(ref[k]).max()), (e, k)
    # go2: 12 friction-loss rows + four condim-6 feet (6 rows + 5 reserved virtual rows each) leave 8 of the 64 slots
    tally.finish(f'elliptic {robot}', min_checked=0.4, max_tie=0.1, max_budget=0.6)
    assert ncon >= nchecked


def test_divergence_guard_freezes_and_flags_the_env():
    """mj_checkAcc analogue: a state whose acceleration is not finite must not propagate NaN - the env keeps a finite state,
    and is flagged terminated + truncated (so that an auto-resetting batch re-spawns it); healthy envs are untouched."""
    mm = marshalled('mini_cheetah', solver=1, iterations=20)
    q = np.tile(mm.md.key_qpos[0], (2, 1)); q[:, 2] = 0.6
    v = np.zeros((2, 18), np.float32); v[0, 8] = np.float32(3e19)      # absurd joint velocity -> Coriolis terms overflow
    st = emu_step(mm, np.zeros((2, 12)), q, v)
    assert st['terminated'].tolist() == [1, 0] and st['truncated'].tolist() == [1, 0]
    assert np.isfinite(st['qpos']).all() and np.isfinite(st['qacc']).all()
    assert np.isfinite(st['qvel'][1]).all() and np.isfinite(st['obs'][1]).all()


def _random_boxes_scene(hip):
    from gym_quadruped_amd.terrain import generate_terrain
    return generate_terrain('random_boxes', hip, seed=10)


@pytest.mark.parametrize('robot', ['aliengo', 'hyqreal1'])   # pyramidal / elliptic cones (go2's condim-6 feet + 27 link geoms
def test_world_boxes_step_matches_oracle(robot):                # exhaust the 64-row budget when it sinks into a box field)
    """BOXES kernel variant on the reference's random_boxes scene (100 tilted boxes, seed 10): rows, frames and the Newton
    solution against the fp64 oracle, robots dropped over the box field; plus the raised-floor identity (a wide slab
    under the robot == the floor plane that much lower, bit-for-bit the same rows)."""
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    hip = get_robot_config(robot).hip_height
    scene, lim = _random_boxes_scene(hip)
    kw = dict(solver=1, iterations=100)
    mm = marshalled(robot, tolerance=1e-8, noise_floor=1e-5, boxes=scene['boxes'], **kw)
    mmN = marshalled(robot, tolerance=1e-13, boxes=scene['boxes'], **kw)
    rng = np.random.default_rng(12)
    n = 16
    qpos, qvel = random_states(mm.md, n, rng, z_range=(0.75 * hip, 1.05 * hip))
    # over the box field (boxes start at x ~ 0.5 + 2 hip, y ~ -3 .. +3 hip-scaled): sample xy inside the spawn limits
    qpos[:, 0] = rng.uniform(0.5 + 2 * hip, 0.5 + 10 * hip, n); qpos[:, 1] = rng.uniform(-3 + 2 * hip, -3 + 12 * hip, n)
    qpos[:, 2] += 0.25 * hip    # the boxes are slabs of height ~hip/2 centred 2 cm above the floor: tops at ~hip/4
    qvel = qvel.astype(np.float32)
    warm = rng.normal(0, 3, (n, 18)).astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), warm=warm.copy(), debug_envs=n)
    o = Oracle(mmN)
    nbox_con = nchecked = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e].astype(np.float64), warm[e].astype(np.float64), np.zeros(18)); o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        nefc = int(dbg(rec, 'nefc')[0])
        if (o.ncon and o.get('contact_tiegap').min() < 3e-6) or nefc != o.nefc:
            continue
        nchecked += 1
        nbox_con += int((np.abs(o.contact_frame[:, 0, 2] - 1.0) > 1e-6).sum()) if o.ncon else 0
        J = dbg(rec, 'efc_J').reshape(64, 18)[:nefc]
        np.testing.assert_allclose(J, o.efc_J, atol=3e-5 * max(1.0, np.abs(o.efc_J).max()))
        np.testing.assert_allclose(dbg(rec, 'efc_R')[:nefc], o.efc_R, rtol=3e-4)
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:nefc], o.efc_aref, atol=3e-4 * max(1.0, np.abs(o.efc_aref).max()))
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-4 * max(1.0, np.abs(o.qacc).max()), e
        assert np.abs(st['qvel'][e] - o.qvel).max() < 5e-4
        ref, t, inv = o.get_obs(ALL_OBS, np.zeros(4))
        got = split_obs(st['obs'][e], ALL_OBS)
        for k in ('contact_forces', 'contact_forces:base', 'contact_state', 'feet_vel'):
            assert np.abs(got[k] - ref[k]).max() < 1e-2 * max(1.0, np.abs(ref[k]).max(), 0.1 * 9.81 * mm.md.total_mass), (e, k)
        assert bool(st['terminated'][e]) == t
    assert nchecked >= n // 2 and nbox_con >= 4, (nchecked, nbox_con)


@pytest.mark.parametrize('scene_name', ['random_boxes', 'flat'])
def test_pgs_with_world_boxes_and_self_collision_matches_oracle_pgs(scene_name):
    """PGS (the solver the north-star names) beyond the flat floor: world boxes + robot self-collision rows (pyramidal cones; aliengo's
    box / capsule link geoms take the exact pair routines).  The collision stages use the L'DL factors' LDS as scratch, so this variant
    factors after the rows are built and carries the contact normals across the solve in a register (gq_step_body.h S8): same rows as
    the Newton variant, and the force / acceleration of the oracle's PGS after the same number of sweeps."""
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    robot = 'aliengo'
    hip = get_robot_config(robot).hip_height
    boxes = _random_boxes_scene(hip)[0]['boxes'] if scene_name == 'random_boxes' else None
    mm = marshalled(robot, solver=0, iterations=40, tolerance=0.0, boxes=boxes, self_collision=True)
    rng = np.random.default_rng(21)
    n = 16
    n = 24 if boxes is not None else 16
    qpos, qvel = random_states(mm.md, n, rng, z_range=(0.85 * hip, 1.15 * hip) if boxes is not None else (0.6 * hip, 1.0 * hip))
    if boxes is not None:   # (feet and a link or two on the boxes: a robot lying in the box field exceeds the 12-contact capacity, a case of its own)
        qpos[:, 0] = rng.uniform(0.5 + 2 * hip, 0.5 + 10 * hip, n); qpos[:, 1] = rng.uniform(-3 + 2 * hip, -3 + 12 * hip, n)
        qpos[:, 2] += 0.25 * hip
    if boxes is None:   # legs folded across each other: states in which the oracle finds robot-robot contacts, inside the row capacity
        qpos, qvel = self_contact_states(mm.md, n, rng, Oracle(marshalled(robot, solver=1, self_collision=True)), z=(0.9 * hip, 1.6 * hip), cone=0, max_over=0.0)
    qvel = qvel.astype(np.float32)
    warm = rng.normal(0, 3, (n, 18)).astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), warm=warm.copy(), debug_envs=n)
    o = Oracle(mm)
    nchecked = nself = nworld = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e].astype(np.float64), warm[e].astype(np.float64), np.zeros(18)); o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        nefc = int(dbg(rec, 'nefc')[0])
        if (o.ncon and o.get('contact_tiegap').min() < 3e-6) or nefc != o.nefc:
            continue
        nchecked += 1
        if o.ncon:
            nworld += int((np.abs(o.contact_frame[:, 0, 2] - 1.0) > 1e-6).sum())
            nself += int((o.get('contact_body1')[:o.ncon] > 0).sum())
        J = dbg(rec, 'efc_J').reshape(64, 18)[:nefc]
        np.testing.assert_allclose(J, o.efc_J, atol=3e-5 * max(1.0, np.abs(o.efc_J).max()))
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:nefc], o.efc_aref, atol=3e-4 * max(1.0, np.abs(o.efc_aref).max()))
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 3e-4 * max(1.0, np.abs(o.qacc).max()), e
        assert np.abs(st['qvel'][e] - o.qvel).max() < 5e-4
        ref, t, inv = o.get_obs(ALL_OBS, np.zeros(4))
        got = split_obs(st['obs'][e], ALL_OBS)
        for k in ('contact_forces', 'contact_forces:base', 'contact_state'):   # S11 reads the contact normals that were parked in a register
            assert np.abs(got[k] - ref[k]).max() < 1e-2 * max(1.0, np.abs(ref[k]).max(), 0.1 * 9.81 * mm.md.total_mass), (e, k)
        assert bool(st['terminated'][e]) == t
    assert nchecked >= n // 2, nchecked
    if boxes is not None:
        assert nworld >= 4, nworld
    else:
        assert nself >= 1, nself


def test_world_box_slab_equals_raised_floor_in_the_kernel():
    from gym_quadruped_amd.terrain import _box
    H = 1.37
    slab = _box([0.0, 0.0, H - 1.0], [0.0, 0.0, 0.0], [12.0, 12.0, 2.0])
    # a hull robot without its hull graphs (see test_flat_height_field_equals_raised_floor_in_the_kernel): one point per geom on the floor - the
    # support vertex - and on a box - the convex routine's single contact, which is that vertex against the slab's top face
    mmF = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-8, mesh_graph=False)
    mmB = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-8, boxes=[slab], mesh_graph=False)
    rng = np.random.default_rng(3)
    n = 8
    qpos, qvel = random_states(mmF.md, n, rng, z_range=(0.18, 0.3))
    qvel = qvel.astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 10).astype(np.float32)
    a = emu_step(mmF, ctrl, qpos.copy(), qvel.copy(), debug_envs=n)
    qb = qpos.copy(); qb[:, 2] += H
    b = emu_step(mmB, ctrl, qb, qvel.copy(), debug_envs=n)
    seen = 0
    for e in range(n):
        na, nb = int(dbg(a['debug'][e], 'nefc')[0]), int(dbg(b['debug'][e], 'nefc')[0])
        assert na == nb
        seen += na > 12
        np.testing.assert_allclose(dbg(b['debug'][e], 'qacc'), dbg(a['debug'][e], 'qacc'), rtol=2e-4, atol=2e-4 * max(1.0, np.abs(dbg(a['debug'][e], 'qacc')).max()))
        np.testing.assert_allclose(b['qvel'][e], a['qvel'][e], atol=1e-4)
    assert seen >= 4


def test_reset_lifts_clear_of_world_boxes():
    """Reset on the random_boxes scene: spawn inside the scene's limits, then the reference's lift loop (:376-388) - here
    re-evaluating floor AND nearby boxes per iteration - must end with no foot-body geom touching anything."""
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    hip = get_robot_config('aliengo').hip_height
    scene, lim = _random_boxes_scene(hip)
    mm = marshalled('aliengo', solver=1, terrain_limits=lim, boxes=scene['boxes'])
    cfg = default_reset_cfg(seed=4242, hip_height=hip)
    n = 12
    st = emu_reset(mm, n, cfg, episode=np.arange(n))
    o = Oracle(mm)
    lifted = nfailed = 0
    for e in range(n):
        assert lim[1] <= st['qpos'][e, 0] <= lim[0] and lim[3] <= st['qpos'][e, 1] <= lim[2]
        o.set_state(st['qpos'][e], st['qvel'][e].astype(np.float64), np.zeros(18), np.zeros(18)); o.forward(np.zeros(12), stage=1)
        bodies = o.get('contact_body') if o.ncon else []
        calf_touch = [c for c, bd in enumerate(bodies) if (int(bd) - 2) % 3 == 2]
        if st['lift_failed'][e]:
            # a foot leaving a box through a steep side face gains only 1.1 |dist| n_z per iteration: after the reference's
            # 100 iterations a hair of penetration is left and the reference raises RuntimeError (:387-388); here: the flag
            nfailed += 1
            assert calf_touch and np.abs(o.get('contact_dist')[calf_touch]).max() < 1e-3
        else:
            assert not calf_touch, 'a calf body still touches the floor or a box'
        lifted += st['qpos'][e, 2] > hip + 1e-4
    assert lifted >= 3 and nfailed <= 2   # some spawn poses did sit in a box


@pytest.mark.parametrize('robot', ['aliengo', 'hyqreal1'])   # pyramidal / elliptic cones
def test_perlin_height_field_step_matches_oracle(robot):
    """BOXES kernel variant on the reference's perlin scene (128 x 128 height field, terrain.py:345-356): contact rows,
    frames and the Newton solution against the fp64 oracle for robots dropped onto the hills."""
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    from gym_quadruped_amd.terrain import generate_terrain
    hip = get_robot_config(robot).hip_height
    scene, lim = generate_terrain('perlin', hip)
    kw = dict(solver=1, iterations=100, hfield=scene['hfield'], terrain_limits=lim)
    mm = marshalled(robot, tolerance=1e-8, noise_floor=1e-5, **kw)
    mmN = marshalled(robot, tolerance=1e-13, **kw)
    rng = np.random.default_rng(21)
    n = 16
    qpos, qvel = random_states(mm.md, n, rng, z_range=(0.75 * hip, 1.05 * hip))
    qpos[:, 0:2] = rng.uniform(-0.7 * lim[0], 0.7 * lim[0], (n, 2))
    # terrain elevation under the base: put the robots at their drawn height above the local surface
    hf = scene['hfield']; data = np.asarray(hf['data'], float); sx, sy, sz, _ = hf['size']
    ci = np.clip(((qpos[:, 0] + sx) / (2 * sx) * (data.shape[1] - 1)).round().astype(int), 0, data.shape[1] - 1)
    ri = np.clip(((qpos[:, 1] + sy) / (2 * sy) * (data.shape[0] - 1)).round().astype(int), 0, data.shape[0] - 1)
    qpos[:, 2] += sz * data[ri, ci]
    qvel = qvel.astype(np.float32)
    warm = rng.normal(0, 3, (n, 18)).astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), warm=warm.copy(), debug_envs=n)
    o = Oracle(mmN)
    nhf_con = nchecked = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e].astype(np.float64), warm[e].astype(np.float64), np.zeros(18)); o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        nefc = int(dbg(rec, 'nefc')[0])
        if (o.ncon and o.get('contact_tiegap').min() < 3e-6) or nefc != o.nefc:
            continue
        nchecked += 1
        nhf_con += int((np.abs(o.contact_frame[:, 0, 2] - 1.0) > 1e-9).sum()) if o.ncon else 0
        J = dbg(rec, 'efc_J').reshape(64, 18)[:nefc]
        np.testing.assert_allclose(J, o.efc_J, atol=3e-5 * max(1.0, np.abs(o.efc_J).max()))
        np.testing.assert_allclose(dbg(rec, 'efc_R')[:nefc], o.efc_R, rtol=3e-4)
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:nefc], o.efc_aref, atol=3e-4 * max(1.0, np.abs(o.efc_aref).max()))
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-4 * max(1.0, np.abs(o.qacc).max()), e
        assert np.abs(st['qvel'][e] - o.qvel).max() < 5e-4
        ref, t, inv = o.get_obs(ALL_OBS, np.zeros(4))
        got = split_obs(st['obs'][e], ALL_OBS)
        for k in ('contact_forces', 'contact_forces:base', 'contact_state', 'feet_vel'):
            assert np.abs(got[k] - ref[k]).max() < 1e-2 * max(1.0, np.abs(ref[k]).max(), 0.1 * 9.81 * mm.md.total_mass), (e, k)
        assert bool(st['terminated'][e]) == t
    assert nchecked >= n // 2 and nhf_con >= 8, (nchecked, nhf_con)


def test_flat_height_field_equals_raised_floor_in_the_kernel():
    H = 1.37
    flat = dict(data=np.zeros((17, 17), np.float32), size=(8.0, 8.0, 1.0, 0.01), pos=(0.0, 0.0, H))
    # a hull robot: its geoms meet the plane and the height field alike in ONE point (the primitive geoms of aliengo take the
    # multi-point plane routines on the floor and the single-point cloud rule on a height field)
    # (mesh_graph off: with its hull graph a mesh brings up to three points to the floor, mjc_PlaneConvex's neighbour walk, and still one to the height field)
    mmF = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-8, mesh_graph=False)
    mmH = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-8, hfield=flat, mesh_graph=False)
    rng = np.random.default_rng(3)
    n = 8
    qpos, qvel = random_states(mmF.md, n, rng, z_range=(0.18, 0.3))
    qvel = qvel.astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 10).astype(np.float32)
    a = emu_step(mmF, ctrl, qpos.copy(), qvel.copy(), debug_envs=n)
    qb = qpos.copy(); qb[:, 2] += H
    b = emu_step(mmH, ctrl, qb, qvel.copy(), debug_envs=n)
    seen = 0
    for e in range(n):
        na, nb = int(dbg(a['debug'][e], 'nefc')[0]), int(dbg(b['debug'][e], 'nefc')[0])
        assert na == nb
        seen += na > 12
        np.testing.assert_allclose(dbg(b['debug'][e], 'qacc'), dbg(a['debug'][e], 'qacc'), rtol=2e-4, atol=2e-4 * max(1.0, np.abs(dbg(a['debug'][e], 'qacc')).max()))
        np.testing.assert_allclose(b['qvel'][e], a['qvel'][e], atol=1e-4)
    assert seen >= 4


def test_reset_lifts_clear_of_the_height_field():
    """Reset on the perlin scene: the keyframe pose spawns inside the hills (elevation up to 2 x hip height); the lift loop
    (:376-388), re-evaluating floor and height field per iteration, must end with no calf-body geom touching anything."""
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    from gym_quadruped_amd.terrain import generate_terrain
    hip = get_robot_config('aliengo').hip_height
    scene, lim = generate_terrain('perlin', hip)
    mm = marshalled('aliengo', solver=1, terrain_limits=lim, hfield=scene['hfield'])
    cfg = default_reset_cfg(seed=77, hip_height=hip)
    n = 12
    st = emu_reset(mm, n, cfg, episode=np.arange(n))
    o = Oracle(mm)
    lifted = 0
    for e in range(n):
        assert lim[1] <= st['qpos'][e, 0] <= lim[0] and lim[3] <= st['qpos'][e, 1] <= lim[2]
        assert not st['lift_failed'][e]
        o.set_state(st['qpos'][e], st['qvel'][e].astype(np.float64), np.zeros(18), np.zeros(18)); o.forward(np.zeros(12), stage=1)
        bodies = o.get('contact_body') if o.ncon else []
        assert not [c for c, bd in enumerate(bodies) if (int(bd) - 2) % 3 == 2], 'a calf body still touches the terrain'
        lifted += st['qpos'][e, 2] > hip + 1e-3
    assert lifted >= n // 2


def test_general_impedance_power_matches_oracle():
    """solimp power other than MuJoCo's default 2 (and than 1): the kernel's exp2/log2 form of x^p / mid^(p-1) against the
    oracle's pow()."""
    import copy
    mm0 = marshalled('aliengo', solver=1, iterations=100, tolerance=1e-8)
    md = copy.copy(mm0.md)
    md.geom_solimp = np.array(md.geom_solimp, dtype=np.float64).reshape(-1, 5).copy()
    md.geom_solimp[:, 4] = 3.5; md.geom_solimp[:, 3] = 0.3
    floor = dict(solimp=(0.9, 0.95, 0.001, 0.3, 3.5))
    from gym_quadruped_amd.cabi import MarshalledModel
    kw = dict(qpos0=np.array(mm0.md.qpos0), feet_geom_names=None, solver=1, iterations=100, floor=floor)
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    kw['feet_geom_names'] = get_robot_config('aliengo').feet_geom_names
    mm = MarshalledModel(md, tolerance=1e-8, **kw)
    mmN = MarshalledModel(md, tolerance=1e-13, **kw)
    rng = np.random.default_rng(5)
    n = 8
    qpos, qvel = random_states(md, n, rng, z_range=(0.3, 0.42))
    qvel = qvel.astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 10).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), debug_envs=n)
    o = Oracle(mmN)
    rows = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e].astype(np.float64), np.zeros(18), np.zeros(18)); o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        nefc = int(dbg(rec, 'nefc')[0])
        assert nefc == o.nefc
        rows += nefc - 12
        np.testing.assert_allclose(dbg(rec, 'efc_R')[:nefc], o.efc_R, rtol=3e-4)
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:nefc], o.efc_aref, atol=3e-4 * max(1.0, np.abs(o.efc_aref).max()))
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-4 * max(1.0, np.abs(o.qacc).max()), e
    assert rows > 40


@pytest.mark.parametrize('robot,want_cross', [('mini_cheetah', True), ('aliengo', False), ('aliengo', True), ('go2', True), ('go1', None), ('b2', True)])
def test_robot_self_collision_matches_oracle(robot, want_cross):
    """Robot-robot contacts (MuJoCo's default contype = conaffinity = 1; capsule proxies, selfcol.py): pair filter, broad
    and narrow phase, two-body Jacobian rows, and the Newton step - tree-sparse when the contact stays inside one leg or
    touches the trunk, dense when it couples two legs - against the oracle: contact list, J, R, aref, forces, qacc."""
    n = 10
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-10, noise_floor=0.0)
    o = Oracle(marshalled(robot, solver=1, iterations=100, tolerance=1e-12))
    rng = np.random.default_rng(7)
    qpos, qvel = self_contact_states(mm.md, n, rng, o, want_cross=want_cross)
    qvel = qvel.astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), debug_envs=n, friction=np.full(n, 0.7, np.float32))
    nself = nchecked = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18), 0.0, 0.7); o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        ne = o.nefc
        if not oracle_fits_self_budget(o, mm.md.cone == 1):
            continue
        if o.get('contact_tiegap').min() < 3e-7:   # a contact whose POINT is not determined (two faces, a face and an edge: gq_oracle.c cvx_point_tie) or a deepest-vertex tie
            continue
        nchecked += 1
        nself += int((o.get('contact_body1') > 0).sum())
        assert int(dbg(rec, 'nefc')[0]) == ne and int(dbg(rec, 'ncon')[0]) == o.ncon, (e, dbg(rec, 'nefc')[0], ne)
        np.testing.assert_allclose(dbg(rec, 'efc_J').reshape(64, 18)[:ne], o.efc_J, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(dbg(rec, 'efc_R')[:ne], o.efc_R, rtol=2e-4)
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:ne], o.efc_aref, rtol=2e-4, atol=5e-2)
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-4 * max(1.0, np.abs(o.qacc).max()), (e, dbg(rec, 'niter'))
        fmax = max(1.0, np.abs(o.efc_force).max())
        assert np.abs(dbg(rec, 'efc_force')[:ne] - o.efc_force).max() < (2e-3 if mm.md.cone == 0 else 2e-2) * fmax
        assert np.abs(st['qvel'][e] - o.qvel).max() < 5e-4 and np.abs(st['qpos'][e] - o.qpos).max() < 2e-6
        # internal forces: no net force on the base dofs from a robot-robot contact row (Newton's third law)
        J = o.efc_J
        b1 = o.get('contact_body1').astype(int); adr = o.get('contact_efc_address').astype(int); dims = o.get('contact_dim').astype(int)
        for c in range(o.ncon):
            if b1[c] > 0:
                nr = dims[c] if mm.md.cone == 1 else (1 if dims[c] == 1 else 2 * (dims[c] - 1))
                assert np.abs(J[adr[c]:adr[c] + nr, :6]).max() < 1e-12
    assert nchecked >= n // 2 and nself > 0


@pytest.mark.parametrize('robot', ['go1', 'spot', 'b2'])
def test_newton_ends_on_captured_hard_states(robot):
    """States captured from benchmark rollouts on the GPU (tools/capture_stuck.py -> tests/golden/newton_stagnation_*.npz)
    on which the fp32 Newton solver ran into the iteration cap (100 iterations = 1.1 ms for the whole launch):
    go1 - cycling between neighbouring fp32 iterates at a gradient of 1e-7 of its starting value (stagnation exit);
    spot - a line search whose Newton trials overshot each other across a kink of phi' and whose unverified last candidate
    raised the cost (bracketed secant, lower bracket end on exhaustion) - qacc was WRONG there;
    b2 - two contacts between different legs in a model without friction-loss rows: first coupling row 0 was read as "no
    coupling" and the tree-sparse solve was used on a Hessian that is not tree-sparse.
    All must end within the order of the fp64 oracle's iteration count, at the oracle's solution."""
    z = np.load(Path(__file__).parent / 'golden' / f'newton_stagnation_{robot}.npz')
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-8)
    o = Oracle(mm)
    n = len(z['qpos'])
    st = emu_step(mm, z['ctrl'].copy(), z['qpos'].copy(), z['qvel'].copy(), warm=z['warm'].copy(), applied=z['applied'].copy(),
                  friction=z['friction'].copy(), debug_envs=n)
    for e in range(n):
        o.set_state(z['qpos'][e], z['qvel'][e].astype(np.float64), z['warm'][e].astype(np.float64), z['applied'][e].astype(np.float64), 0.0, float(z['friction'][e]))
        o.step(z['ctrl'][e].astype(np.float64))
        nit = int(dbg(st['debug'][e], 'niter')[0])
        assert nit <= 20, (e, nit, o.solver_niter)
        qa = np.array(o.qacc)
        assert np.abs(st['qacc'][e] - qa).max() <= 2e-5 * max(1.0, np.abs(qa).max()), (e, nit)   # spot (condim 6, impratio 100): 8e-6; the others below 1e-6


def test_newton_with_22_virtual_rows_go1_on_boxes():
    """go1 on random_boxes, states captured on the GPU (tools/capture_stuck.py) where the solver needed 31-100 iterations against
    the oracle's 4-12: four condim-6 feet contacts and one condim-3 contact, all in the middle zone of their elliptic cones,
    make 4 x 5 + 2 = 22 virtual Hessian rows = 396 outputs, and the lane-parallel construction held 6 x 64 = 384 - the last row
    kept stale entries, the Hessian was wrong and Newton fell back to the linear rate of a line-searched gradient method.
    Now 8 per lane.  Each state must end within a few iterations, at the oracle's solution when the contact sets agree."""
    from gym_quadruped_amd.terrain import generate_terrain
    from gym_quadruped_amd.robot_cfgs import get_robot_config
    z = np.load(Path(__file__).parent / 'golden' / 'newton_stagnation_go1_random_boxes.npz')
    scene, lim = generate_terrain('random_boxes', get_robot_config('go1').hip_height)
    mm = marshalled('go1', solver=1, iterations=100, tolerance=1e-8, noise_floor=1e-5, boxes=scene['boxes'], terrain_limits=lim)
    o = Oracle(marshalled('go1', solver=1, iterations=100, tolerance=1e-10, boxes=scene['boxes'], terrain_limits=lim))
    n = len(z['qpos'])
    st = emu_step(mm, z['ctrl'].copy(), z['qpos'].copy(), z['qvel'].copy(), warm=z['warm'].copy(), applied=z['applied'].copy(),
                  friction=z['friction'].copy(), debug_envs=n)
    same = 0
    for e in range(n):
        nit = int(dbg(st['debug'][e], 'niter')[0])
        assert nit <= 12, (e, nit, int(z['niter_before_fix'][e]))
        o.set_state(z['qpos'][e], z['qvel'][e].astype(np.float64), z['warm'][e].astype(np.float64), z['applied'][e].astype(np.float64), 0.0, float(z['friction'][e]))
        o.step(z['ctrl'][e].astype(np.float64))
        if int(o.nefc) == int(dbg(st['debug'][e], 'nefc')[0]):   # (the others exceed the kernel's row budget: prefix rule, other tests)
            qa = np.array(o.qacc)
            assert np.abs(st['qacc'][e] - qa).max() <= 5e-5 * max(1.0, np.abs(qa).max()), (e, nit)
            same += 1
    assert same >= 1


@pytest.mark.parametrize('robot', ['aliengo', 'go1', 'b2', 'hyqreal2', 'go2'])
def test_plane_multipoint_contacts_match_oracle(robot):
    """MuJoCo's multi-point plane routines in the kernel (csrc/gq_step_body.h floor_candidates: box corners, both capsule end
    spheres with the axis-aligned frame, the cylinder's four rim points) against the oracle's restatement on poses that put
    trunk, hip and thigh geoms on the floor: contact list (order, distances), rows, frames (through J) and the Newton solution.
    States whose oracle row count exceeds one wavefront's budget are held to the PREFIX rule: the kernel's rows equal the
    oracle's first rows."""
    from test_oracle_invariants import _lying_states
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-8, noise_floor=1e-5, self_collision=False)
    mmN = marshalled(robot, solver=1, iterations=200, tolerance=1e-13, self_collision=False)
    md, o = mm.md, Oracle(mmN)
    rng = np.random.default_rng(23)
    hip = float(mm.desc.key_qpos[2])
    cone = bool(md.cone)
    # keep poses with a multi-point contact of a primitive geom, up to n of them
    n, Q = 20, []
    for q in _lying_states(md, 4000, rng, (0.25 * hip, 0.9 * hip)):
        o.set_state(q, np.zeros(18), np.zeros(18), np.zeros(18)); o.forward(np.zeros(12), stage=1)
        if not o.ncon or o.ncon > 10:
            continue
        g = o.get('contact_geom').astype(int)
        if max(np.bincount(g)) >= 2:
            Q.append(q)
        if len(Q) == n:
            break
    assert len(Q) == n
    qpos = np.stack(Q)
    qvel = rng.normal(0, 0.5, (n, 18)).astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 10).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), debug_envs=n)
    tally = ParityTally(cone=cone, tie_threshold=3e-7)
    types = set()
    for e in range(n):
        o.set_state(qpos[e], qvel[e].astype(np.float64), np.zeros(18), np.zeros(18)); o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        nefc, ncon = int(dbg(rec, 'nefc')[0]), int(dbg(rec, 'ncon')[0])
        cls = tally.classify(e, o, nefc)
        if cls in ('tie', 'mismatch'):
            continue
        # contact list and rows: all of them ('ok') or the kernel's prefix of the oracle's list ('budget')
        og = o.get('contact_geom').astype(int)
        np.testing.assert_allclose(dbg(rec, 'contact_dist')[:ncon], o.get('contact_dist')[:ncon], atol=2e-6)
        J = dbg(rec, 'efc_J').reshape(64, 18)[:nefc]
        np.testing.assert_allclose(J, o.efc_J[:nefc], atol=3e-5 * max(1.0, np.abs(o.efc_J[:nefc]).max()))
        np.testing.assert_allclose(dbg(rec, 'efc_R')[:nefc], o.efc_R[:nefc], rtol=3e-4)
        np.testing.assert_allclose(dbg(rec, 'efc_aref')[:nefc], o.efc_aref[:nefc], atol=3e-4 * max(1.0, np.abs(o.efc_aref[:nefc]).max()))
        ref, t, inv = o.get_obs(ALL_OBS, np.zeros(4))
        assert bool(st['terminated'][e]) == t and bool(st['invalid'][e]) == inv   # body-level test, taken before any capping
        if cls != 'ok':
            continue
        types |= {int(md.geom_type[g]) for g in og if np.count_nonzero(og == g) > 1}
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-4 * max(1.0, np.abs(o.qacc).max()), (e, dbg(rec, 'niter'))
        fmax = max(1.0, np.abs(o.efc_force).max())
        assert np.abs(dbg(rec, 'efc_force')[:nefc] - o.efc_force).max() < 2e-3 * fmax
        assert np.abs(st['qvel'][e] - o.qvel).max() < 5e-4 and np.abs(st['qpos'][e] - o.qpos).max() < 2e-6
        got = split_obs(st['obs'][e], ALL_OBS)
        for k in ('contact_forces', 'contact_forces:base', 'contact_state'):
            assert np.abs(got[k] - ref[k]).max() < 2e-3 * max(1.0, np.abs(ref[k]).max(), 0.05 * 9.81 * md.total_mass), (e, k)
    tally.finish(f'plane multi-point {robot}', min_checked=0.4, max_tie=0.1, max_budget=0.6)
    have = {int(t) for g, t in enumerate(md.geom_type) if md.geom_bodyid[g] != 0 and md.geom_cloudid[g] >= 0 and t in (3, 5, 6)}
    assert types == have, (types, have)   # every primitive type of the robot produced a multi-point contact that was compared


def test_pair_routines_kernel_equals_oracle():
    """csrc/gq_pairs.h (capsule_box, box_box: fp32, called directly under the emulator) against the oracle's restatement
    (oracle/gq_oracle.c, fp64) on random and on resting configurations: same number of points, same order, distances /
    positions / normals to fp32 accuracy.  The oracle versions are pinned against brute-force geometry in
    tests/test_oracle_invariants.py."""
    import ctypes as C
    from scipy.spatial.transform import Rotation
    from helpers import emu_lib
    from test_oracle_invariants import _pair_lib, _np_ptr
    Lo, Le = _pair_lib(), emu_lib()
    P = C.c_void_p
    Le.emu_capsule_box.argtypes = [P, P, C.c_float, P, P, P, C.c_float, P]
    Le.emu_box_box.argtypes = [P, P, P, P, P, P, C.c_float, P]
    rng = np.random.default_rng(12)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    ncap = nbox = multi = 0
    for trial in range(1500):
        h = rng.uniform(0.02, 0.3, 3); R = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix(); c = rng.uniform(-1, 1, 3)
        margin = 0.001
        if trial % 2 == 0:   # capsule - box
            r = rng.uniform(0.005, 0.05)
            if trial % 8 == 0:
                a = np.array([rng.uniform(-h[0], h[0]), rng.uniform(-h[1], h[1]), h[2] + r + rng.uniform(-0.002, 0.0005)])
                b = np.array([rng.uniform(-h[0], h[0]), rng.uniform(-h[1], h[1]), a[2] + rng.uniform(-2e-4, 2e-4)])
                p0, p1 = c + R @ a, c + R @ b
            else:
                p0 = c + R @ (rng.uniform(-1.3, 1.3, 3) * h); p1 = p0 + rng.normal(0, 0.1, 3)
            # fp32 inputs for both, so that the comparison is about the arithmetic only
            p0, p1, cc, Rc, hc = (f32(x).astype(np.float64) for x in (p0, p1, c, R, h))
            rr = float(np.float32(r))
            oo, oe = np.zeros(28), np.zeros(28, np.float32)
            Rc = np.ascontiguousarray(Rc)
            fa = [f32(x) for x in (p0, p1, cc, Rc, hc)]   # keep the fp32 copies alive across the call
            no = Lo.gqo_test_capsule_box(_np_ptr(p0), _np_ptr(p1), rr, _np_ptr(cc), _np_ptr(Rc), _np_ptr(hc), margin, _np_ptr(oo))
            ne = Le.emu_capsule_box(_np_ptr(fa[0]), _np_ptr(fa[1]), rr, _np_ptr(fa[2]), _np_ptr(fa[3]), _np_ptr(fa[4]), margin, _np_ptr(oe))
            ncap += no > 0
        else:
            hb = rng.uniform(0.02, 0.3, 3)
            if trial % 6 == 1:
                hb[:2] = rng.uniform(0.2, 0.9, 2) * h[:2]
                Rb = R @ Rotation.from_euler('z', rng.uniform(-0.3, 0.3)).as_matrix()
                cb = c + R @ np.array([*(rng.uniform(-0.05, 0.05, 2) * h[:2]), h[2] + hb[2] + rng.uniform(-0.003, 0.0008)])
            else:
                Rb = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix()
                cb = c + rng.normal(0, 1, 3) * (h + hb) * 0.8
            cc, Rc, hc, cb, Rb, hb = (f32(x).astype(np.float64) for x in (c, R, h, cb, Rb, hb))
            oo, oe = np.zeros(28), np.zeros(28, np.float32)
            Rc, Rb = np.ascontiguousarray(Rc), np.ascontiguousarray(Rb)
            fa = [f32(x) for x in (cc, Rc, hc, cb, Rb, hb)]
            no = Lo.gqo_test_box_box(_np_ptr(cc), _np_ptr(Rc), _np_ptr(hc), _np_ptr(cb), _np_ptr(Rb), _np_ptr(hb), margin, _np_ptr(oo))
            ne = Le.emu_box_box(_np_ptr(fa[0]), _np_ptr(fa[1]), _np_ptr(fa[2]), _np_ptr(fa[3]), _np_ptr(fa[4]), _np_ptr(fa[5]), margin, _np_ptr(oe))
            nbox += no > 0
        if no and abs(oo[0::7][:no]).max() > 0.02:
            continue   # centimetres of overlap: decisions between nearly equal axes / deepest samples may differ; contacts are created at the margin
        assert no == ne, (trial, no, ne, oo[:7], oe[:7])
        multi += no > 1
        for q in range(no):
            assert abs(oo[7 * q] - oe[7 * q]) < 2e-6, (trial, q, oo[7 * q:7 * q + 7], oe[7 * q:7 * q + 7])
            np.testing.assert_allclose(oe[7 * q + 1:7 * q + 4], oo[7 * q + 1:7 * q + 4], atol=5e-6, err_msg=f'trial {trial} point {q} pos')
            np.testing.assert_allclose(oe[7 * q + 4:7 * q + 7], oo[7 * q + 4:7 * q + 7], atol=2e-4, err_msg=f'trial {trial} point {q} normal')
    assert ncap > 100 and nbox > 100 and multi > 60, (ncap, nbox, multi)


@pytest.mark.parametrize('robot', [None, 'mini_cheetah', 'hyqreal1', 'go1'])
def test_convex_routine_kernel_equals_oracle(robot):
    """csrc/gq_convex.h (GJK + EPA on one wavefront: lane = vertex support scans, lane = face polytope; fp32 points, fp64 simplex and
    face-plane arithmetic) called directly under the emulator against the oracle's restatement (oracle/gq_convex.h, fp64, pinned
    against the exact Minkowski-difference hull in tests/test_oracle_invariants.py): random polytopes and the robots' own mesh /
    cylinder clouds, against an analytic box and against each other, from 12 mm apart to 30 mm deep, with and without an inflation
    radius.  Same contacts; distance to 1e-6 m, normal to 0.1 degree, point to 2e-5 m wherever the oracle says the point is determined
    (not face on face / edge in face) and the polytope did not run into the iteration cap both sides share."""
    import ctypes as C
    from scipy.spatial.transform import Rotation as Rot
    from helpers import emu_lib
    from test_oracle_invariants import _box_corners, _support, convex_oracle
    Le = emu_lib()
    P = C.c_void_p
    Le.emu_convex.argtypes = [P, C.c_int, P, P, P, C.c_float] * 2 + [C.c_float, P]

    def emu(VA, hA, RA, tA, rA, VB, hB, RB, tB, rB, margin):
        arrs = [None if x is None else np.ascontiguousarray(x, dtype=np.float32) for x in (VA, hA, RA, tA, VB, hB, RB, tB)]
        p = [None if a is None else a.ctypes.data_as(P) for a in arrs]
        out = np.zeros(7, np.float32)
        rc = Le.emu_convex(p[0], 0 if arrs[0] is None else len(arrs[0]), p[1], p[2], p[3], rA, p[4], 0 if arrs[4] is None else len(arrs[4]), p[5], p[6], p[7], rB, margin,
                           out.ctypes.data_as(P))
        return rc, float(out[0]), out[1:4].astype(float), out[4:7].astype(float)

    rng = np.random.default_rng(0)
    md = marshalled(robot, solver=1).md if robot else None
    clouds = [c for c in range(len(md.cloud_vertnum)) if md.cloud_vertnum[c] >= 8] if md else None
    cloud = lambda c: md.vert_pos[md.cloud_vertadr[c]:md.cloud_vertadr[c] + md.cloud_vertnum[c]]
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)   # both sides see the same (fp32) inputs
    n = capped = deep = 0
    for trial in range(240 if robot is None else 120):
        if md is None:
            VA = rng.normal(size=(rng.integers(4, 60), 3)) * rng.uniform(0.02, 0.15, size=3)
            VB, hB = (None, rng.uniform(0.05, 0.5, size=3)) if trial % 2 == 0 else (rng.normal(size=(rng.integers(4, 60), 3)) * rng.uniform(0.02, 0.15, size=3), None)
        else:
            VA = cloud(clouds[trial % len(clouds)])
            VB, hB = (None, rng.uniform(0.1, 0.6, size=3)) if trial % 2 == 0 else (cloud(clouds[int(rng.integers(len(clouds)))]), None)
        VA = f32(VA); VB = None if VB is None else f32(VB); hB = None if hB is None else f32(hB)
        RA, RB = (f32(Rot.random(random_state=int(rng.integers(1 << 30))).as_matrix()) for _ in range(2))
        WA, WB0 = VA @ RA.T, (_box_corners(hB) if VB is None else VB) @ RB.T
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        want = rng.uniform(-0.03, 0.012)
        tB = u * (_support(WA, u) + _support(WB0, -u) + 0.05)
        rc0, d0, _, n0, _, _ = convex_oracle(VA, None, RA, np.zeros(3), 0.0, VB, hB, RB, tB, 0.0, 10.0)
        tB = f32(tB - (d0 - want) * n0)
        rA = float(np.float32(rng.choice([0.0, 0.01])))
        rc, dist, pos, nrm, git, eit = convex_oracle(VA, None, RA, np.zeros(3), rA, VB, hB, RB, tB, 0.0, 0.01)
        rk, dk, pk, nk = emu(VA, None, RA, np.zeros(3), rA, VB, hB, RB, tB, 0.0, 0.01)
        if rc != rk:
            assert rc and abs(dist - 0.01) < 2e-6, (trial, rc, rk, dist)   # only a pair AT the margin may be seen by one side alone
            continue
        if not rc:
            continue
        n += 1; deep += dist < -1e-3
        if eit >= 24:   # the shared iteration cap: both sides report the state of an unfinished iteration, which round-off steers
            capped += 1
            assert abs(dk - dist) < 1e-4
            continue
        assert abs(dk - dist) < 1e-6, (trial, dk, dist)
        assert np.degrees(np.arccos(np.clip(nk @ nrm, -1, 1))) < 0.1, (trial, nk, nrm)
        # the point: compared where it is determined (support sets along the normal: not two faces, a face and an edge, parallel edges)
        WB = WB0 + tB
        da = int((WA @ nrm > (WA @ nrm).max() - 1e-6).sum()); db = int((WB @ -nrm > (WB @ -nrm).max() - 1e-6).sum())
        if min(da, db) == 1 or (da == 2 and db == 2):
            assert np.linalg.norm(pk - pos) < 2e-5, (trial, pk, pos, da, db)
    assert n >= 80 and deep >= 30 and capped <= 0.05 * n, (n, deep, capped)


def test_capsule_proxy_mode_matches_oracle():
    """QuadrupedEnv(self_collision='capsule') / GqModelDesc.self_convex = 0: robot-robot pairs that involve a mesh go through the capsule proxies
    of geom_capsule instead of the convex routine, in kernel and oracle alike (the approximate, fast mode: DESIGN.md section 3)."""
    n = 10
    mm = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-10, noise_floor=0.0, self_collision='capsule')
    o = Oracle(marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-12, self_collision='capsule'))
    assert mm.desc.self_convex == 0 and mm.self_collision == 'capsule'
    rng = np.random.default_rng(7)
    qpos, qvel = self_contact_states(mm.md, n, rng, o, want_cross=True)
    qvel = qvel.astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    st = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), debug_envs=n, friction=np.full(n, 0.7, np.float32))
    nchecked = 0
    for e in range(n):
        o.set_state(qpos[e], qvel[e], np.zeros(18), np.zeros(18), 0.0, 0.7); o.step(ctrl[e].astype(np.float64))
        rec = st['debug'][e]
        if not oracle_fits_self_budget(o, False) or o.get('contact_tiegap').min() < 3e-7:
            continue
        nchecked += 1
        ne = o.nefc
        assert int(dbg(rec, 'nefc')[0]) == ne and int(dbg(rec, 'ncon')[0]) == o.ncon
        np.testing.assert_allclose(dbg(rec, 'efc_J').reshape(64, 18)[:ne], o.efc_J, rtol=2e-4, atol=2e-5)
        assert np.abs(dbg(rec, 'qacc') - o.qacc).max() < 2e-4 * max(1.0, np.abs(o.qacc).max())
    assert nchecked >= n // 2


@pytest.mark.parametrize('robot', ['mini_cheetah', 'hyqreal1'])
def test_pair_exchange_changes_nothing(robot):
    """The convex pair exchange (csrc/gq_exchange.h): an env with several hull pairs in reach probes them, publishes all but the first
    undecided one to the batch's queue, claims items back and collects the results - the step it produces is bit-identical to the one
    without the exchange, and items do pass through the queue.  (Under emulation the wavefronts run one after the other: the owner claims
    its own items - every queue operation, no concurrency; the device test in test_gpu_parity.py covers that.)"""
    from helpers import emu_lib
    n = 12
    mm = marshalled(robot, solver=1, iterations=100, tolerance=1e-10, noise_floor=0.0)
    o = Oracle(marshalled(robot, solver=1, iterations=100, tolerance=1e-12))
    rng = np.random.default_rng(11)
    qpos, qvel = self_contact_states(mm.md, n, rng, o, want_cross=True)
    # two envs in the key posture in front (one per window of the emulator's 256-slot table): envs without convex work of their own, whose
    # passing by is what makes the entangled ones publish (gq_exchange.h: owners publish only while potential helpers are around)
    q0, v0 = random_states(mm.md, 2, rng, z_range=(0.6, 0.7))
    q0[:, 7:] = mm.md.key_qpos[0][7:]; q0[:, 3:7] = [1, 0, 0, 0]
    qpos, qvel = np.concatenate([q0, qpos]), np.concatenate([0 * v0, qvel])
    n += 2
    qvel = qvel.astype(np.float32)
    ctrl = (rng.normal(0, 1, (n, 12)) * 20).astype(np.float32)
    L = emu_lib()
    L.emu_set_exchange(0)
    a = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), debug_envs=n)
    stats0 = (C.c_int * 4)(); L.emu_exchange_stats(stats0)
    L.emu_set_exchange(1)
    try:
        b = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), debug_envs=n)
        c = emu_step(mm, ctrl, qpos.copy(), qvel.copy(), debug_envs=n)   # a second launch: the last leaver rewound the queue, new epoch
    finally:
        L.emu_set_exchange(0)
    stats = (C.c_int * 4)(); L.emu_exchange_stats(stats)
    assert stats[0] - stats0[0] > 0, 'no env published a pair: the test does not reach the exchange'
    for k in ('qpos', 'qvel', 'warm', 'obs'):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k
    for e in range(n):
        for f in ('ncon', 'nefc', 'efc_J', 'efc_aref', 'contact_dist', 'contact_geom', 'qacc'):
            assert np.array_equal(dbg(a['debug'][e], f), dbg(b['debug'][e], f)), (e, f)
