"""GPU side of reset() and of the in-episode resampling, through the C-ABI:

* reset_kernel output against the documented Philox draw table (tests/philox_ref.py) and the oracle's lift condition -
  what tests/test_kernel_emulated.py::test_reset_kernel_stream_lift_and_bookkeeping checks on the CPU emulator;
* the reset's own mj_step (quadruped_env.py:397) against the oracle stepping the state the reset kernel wrote;
* '...+reset' command redraw and external_disturbances_kwargs type 'reset' (quadruped_env.py:292-305, :1046-1139) in the
  step kernel's epilogue: countdown, redraw values from the documented stream, "acts from the next step on" (quirk B9);
* the benchmarked kernel (Newton, next-step auto-reset) at the headline size: 4096-env mini_cheetah property test;
* the advisor's round-1 findings: permuted legs_order in feet_contact_state, IMU bias in state_dict, info as a plain dict.
"""
import numpy as np
import pytest
import torch

from helpers import ALL_OBS, marshalled
from philox_ref import draws, resample_draws

pytestmark = pytest.mark.gpu


def _env(robot='mini_cheetah', n=64, **kw):
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    kw.setdefault('state_obs_names', ('qpos', 'qvel'))
    return QuadrupedEnv(robot, num_envs=n, device='cuda:0', **kw)


@pytest.mark.parametrize('robot', ['mini_cheetah', 'go2'])
def test_reset_matches_draw_table_lift_rule_and_oracle_step(robot):
    """gq_reset = reset_kernel (state write from the Philox draw table, lift loop) + one masked mj_step.  The state the
    kernel wrote is reconstructed from the table up to the lift (z only), the lift is checked against the reference's stop
    condition with the oracle's mj_step1 (no calf-body contact, and - the rule raises by 1.1 max|dist| per iteration - not
    higher than needed), and the post-reset state must equal the oracle's mj_step of the reconstructed state."""
    from oracle.oracle import Oracle
    n, seed, off = 96, 987654321987, 1000
    env = _env(robot, n, seed=seed, env_id_offset=off, base_vel_command_type='random+rotate', ref_base_lin_vel=(0.5, 1.0),
               ref_base_ang_vel=(-0.3, 0.3), ground_friction_coeff=(0.2, 1.5), state_obs_names=tuple(ALL_OBS))
    lim = env.terrain_limits
    obs = env.reset(random=True)
    torch.cuda.synchronize()
    assert not bool(env.lift_failed.any())
    qp, qv = env.qpos.cpu().numpy(), env.qvel.cpu().numpy()
    cmd, fr = env._cmd.cpu().numpy(), env._friction.cpu().numpy()
    assert env._episode.cpu().numpy().tolist() == [1] * n and env._step_num.cpu().numpy().tolist() == [0] * n
    assert np.allclose(env._time.cpu().numpy(), 0.002) and not env._applied.cpu().numpy().any()
    key = np.asarray(env.mjModel.key_qpos[0])
    hip = float(env.robot_cfg.hip_height)
    o = Oracle(marshalled(robot, solver=1, iterations=100, tolerance=1e-12, terrain_limits=lim))
    amp = np.float32(20 * np.pi / 180)
    from scipy.spatial.transform import Rotation
    nlift = 0
    for e in range(n):
        u = draws(seed, off + e, 0)
        # ---- the state reset_kernel wrote, from the table (reference :343-373)
        q0 = key.copy()
        q0[7:] = key[7:] + (2 * u[0:12] - 1) * amp
        v0 = np.zeros(18); v0[6:] = (2 * u[12:24] - 1) * 0.5
        x = lim[0] + (lim[1] - lim[0]) * float(u[24]); y = lim[2] + (lim[3] - lim[2]) * float(u[25])
        roll, pitch = (2 * u[26] - 1) * np.float32(10 * np.pi / 180), (2 * u[27] - 1) * np.float32(10 * np.pi / 180)
        yaw = np.arctan2(-y, -x)
        q0[0], q0[1] = x, y
        q0[3:7] = Rotation.from_euler('xyz', [roll, pitch, yaw]).as_quat(scalar_first=True)
        # command and friction (reference :400-404)
        norm, head = 0.5 + 0.5 * u[28], (2 * u[29] - 1) * np.pi
        np.testing.assert_allclose(cmd[e], [norm * np.cos(head), norm * np.sin(head), 0, -0.3 + 0.6 * u[30]], atol=2e-6)
        assert abs(fr[e] - (0.2 + 1.3 * u[31])) < 1e-6
        # ---- lift: the kernel's z is the only unknown; recover it from the post-step state by undoing the oracle step
        # (search z such that oracle(q0 with z).step() lands on the GPU state: z enters linearly while nothing touches)
        o.set_state(np.r_[q0[:2], hip, q0[3:]], v0, np.zeros(18), np.zeros(18), 0.0, -1.0)
        o.forward(np.zeros(12), stage=1)
        bodies = o.get('contact_body').astype(int) if o.ncon else np.zeros(0, int)
        dist = o.get('contact_dist') if o.ncon else np.zeros(0)
        calf = np.array([(b - 2) % 3 == 2 for b in bodies], bool)
        z = hip
        it = 0
        while calf.any() and it < 100:   # the reference's loop (:378-388) on the oracle
            z += 1.1 * np.abs(dist[calf]).max()
            o.set_state(np.r_[q0[:2], z, q0[3:]], v0, np.zeros(18), np.zeros(18), 0.0, -1.0)
            o.forward(np.zeros(12), stage=1)
            bodies = o.get('contact_body').astype(int) if o.ncon else np.zeros(0, int)
            dist = o.get('contact_dist') if o.ncon else np.zeros(0)
            calf = np.array([(b - 2) % 3 == 2 for b in bodies], bool)
            it += 1
        nlift += it > 0
        q0[2] = z
        # ---- the reset's own mj_step with zero control and the XML frictions (:397, friction committed after it)
        o.set_state(q0, v0, np.zeros(18), np.zeros(18), 0.0, -1.0)
        o.step(np.zeros(12))
        assert np.abs(qp[e] - o.qpos).max() < 2e-6 + 1e-9 * np.abs(o.qpos).max(), (e, it, qp[e] - o.qpos)
        assert np.abs(qv[e] - o.qvel).max() < 2e-4, (e, it)
        ref, t, inv = o.get_obs(ALL_OBS, cmd[e])
        for k in ('feet_pos:base', 'base_lin_vel:base', 'gravity_vector:base'):
            assert np.abs(obs[k][e].cpu().numpy() - ref[k]).max() < 1e-4 * max(1.0, np.abs(ref[k]).max()), (e, k)
    assert nlift > 0, 'no env needed the lift loop: the test does not exercise it'
    # a second reset draws from episode 1, not from the same block
    env.reset(random=True)
    torch.cuda.synchronize()
    u = draws(seed, off + 3, 1)
    np.testing.assert_allclose(env.qpos[3, 7:].cpu().numpy(), key[7:] + (2 * u[0:12] - 1) * amp, atol=2e-3)  # one step of motion
    assert abs(float(env._friction[3]) - (0.2 + 1.3 * u[31])) < 1e-6


def test_command_redraw_and_disturbance_resampling_in_the_step_epilogue():
    """'forward+rotate+reset' commands and a 'reset'-type disturbance wrench, resampled by the step kernel (no host ops):
    countdown semantics of quadruped_env.py:292-305, redraw values from the documented Philox stream, wrench written to
    qfrc_applied[:6] at the END of a step so that it acts from the next one on (quirk B9)."""
    n, seed = 48, 4242
    kw = dict(base_vel_command_type='random+rotate+reset', ref_base_lin_vel=(0.3, 0.9), ref_base_ang_vel=(-0.4, 0.4),
              external_disturbances_kwargs={'type': 'reset', 'x': (-20.0, 20.0), 'z': (5.0,), 'yaw': (-3.0, 3.0)}, seed=seed,
              solver='newton')
    env = _env('aliengo', n, **kw)
    env.reset(random=True)
    torch.cuda.synchronize()
    h = env._h9.cpu().numpy()
    # reset restarted the command interval with a draw in [1000, 2999] from the reset table (draw 32)
    for e in range(n):
        assert h[e, 0] == 0 and h[e, 1] == 1000 + int(np.float32(2000.0) * draws(seed, e, 0)[32]) and 1000 <= h[e, 1] <= 2999
    assert np.all((h[:, 4] >= 1000) & (h[:, 4] <= 2999)) and np.all(h[:, 3] == 0)
    ext0 = env._ext_dist.cpu().numpy().copy()
    assert np.all(ext0[:, 2] == 5.0) and np.all(ext0[:, 1] == 0) and np.all(np.abs(ext0[:, 0]) <= 20) and ext0[:, 0].std() > 1
    assert not env._applied.cpu().numpy().any()        # reset zeroes qfrc_applied (:335); nothing applied yet
    # shorten the countdowns: env e redraws its command at step 2 + e % 3, its wrench at step 3 + e % 2
    env._h9[:, 1] = torch.as_tensor(2 + np.arange(n) % 3, dtype=torch.int32, device='cuda:0')
    env._h9[:, 4] = torch.as_tensor(3 + np.arange(n) % 2, dtype=torch.int32, device='cuda:0')
    cmd0 = env._cmd.cpu().numpy().copy()
    twin = _env('aliengo', n, seed=seed, solver='newton', base_vel_command_type='random+rotate')   # no disturbances
    twin.reset(random=True)
    assert torch.equal(env.qpos, twin.qpos)
    act = torch.zeros(n, 12, device='cuda:0')
    for step in range(1, 6):
        env.step(act); twin.step(act)
        torch.cuda.synchronize()
        h, cmd, ext, app = env._h9.cpu().numpy(), env._cmd.cpu().numpy(), env._ext_dist.cpu().numpy(), env._applied.cpu().numpy()
        for e in range(n):
            t_cmd, t_dist = 2 + e % 3, 3 + e % 2
            if step < t_cmd:
                assert h[e, 0] == step and h[e, 2] == 0 and np.array_equal(cmd[e], cmd0[e])
            else:
                u = resample_draws(seed, e, 0, 'cmd')
                norm, head, yd = 0.3 + 0.6 * u[0], (2 * u[1] - 1) * np.pi, -0.4 + 0.8 * u[2]
                np.testing.assert_allclose(cmd[e], [norm * np.cos(head), norm * np.sin(head), 0, yd], atol=2e-6)
                assert h[e, 2] == 1 and h[e, 0] == step - t_cmd and h[e, 1] == 1000 + int(np.float32(2000.0) * u[3])
            if step < t_dist:
                assert h[e, 3] == step and np.array_equal(ext[e], ext0[e])
            else:
                u = resample_draws(seed, e, 0, 'dist')
                want = [-20 + 40 * u[0], 0, 5.0, 0, 0, -3 + 6 * u[5]]
                np.testing.assert_allclose(ext[e], want, atol=1e-5)
                assert h[e, 5] == 1 and h[e, 3] == step - t_dist and h[e, 4] == 1000 + int(np.float32(2000.0) * u[6])
            assert np.array_equal(app[e, :6], ext[e]) and not app[e, 6:].any()      # :305
        if step == 1:   # the wrench was written at the END of step 1: the first step is identical to the undisturbed twin
            assert torch.equal(env.qpos, twin.qpos) and torch.equal(env.qvel, twin.qvel)
    # from step 2 on the 5 N lift and the pushes act
    assert not torch.equal(env.qvel, twin.qvel) and float((env.qvel[:, 2] - twin.qvel[:, 2]).mean()) > 0.0
    # oracle cross-check of one disturbed step: qfrc_applied enters qfrc_smooth
    from oracle.oracle import Oracle
    o = Oracle(marshalled('aliengo', solver=1, iterations=100, tolerance=1e-12))
    q0, v0, w0, a0 = env.qpos.cpu().numpy().copy(), env.qvel.cpu().numpy().copy(), env._warm.cpu().numpy().copy(), env._applied.cpu().numpy().copy()
    fr = env._friction.cpu().numpy().copy()
    env.step(act)
    torch.cuda.synchronize()
    for e in range(0, n, 5):
        o.set_state(q0[e], v0[e], w0[e], a0[e], 0.0, float(fr[e])); o.step(np.zeros(12))
        assert np.abs(env.qvel[e].cpu().numpy() - o.qvel).max() < 5e-4


def test_next_step_auto_reset_restarts_the_command_interval():
    n = 512
    env = _env('mini_cheetah', n, base_vel_command_type='forward+reset', auto_reset='next_step', seed=3, solver='newton')
    env.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(0)
    prev_term = torch.zeros(n, dtype=torch.bool, device='cuda:0')
    prev_after = env._h9[:, 0].clone()
    nres = 0
    for _ in range(80):
        _, _, term, _, _ = env.step(torch.randn(n, 12, generator=g, device='cuda:0') * 50)
        after = env._h9[:, 0]
        # an env that terminated last step spent this one on reset(): interval restarted (:1068-1070), no increment
        assert bool((after[prev_term] == 0).all()) and bool((after[~prev_term] == prev_after[~prev_term] + 1).all())
        nres += int(prev_term.sum())
        prev_term, prev_after = term.clone(), after.clone()
    assert nres > 0


def test_headline_kernel_4096_envs_newton_properties():
    """BASELINE config 2 on the kernel the benchmark times (mini_cheetah, flat, 4096 envs, Newton, next-step auto-reset,
    50 N(0,1) torques, ALL_OBS): finite state, unit quaternions, feet above the soft-contact depth, unilateral bounded
    contact forces, contact_state consistent with the forces, envs terminate and are re-spawned, step counters restart."""
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    n = 4096
    env = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, solver='newton',
                       solver_iterations=100, solver_tolerance=1e-8, auto_reset='next_step', seed=1000)
    env.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(0)
    nterm = 0
    for _ in range(300):
        obs, rew, term, trunc, info = env.step(torch.randn(n, 12, generator=g, device='cuda:0') * 50)
        nterm += int(term.sum())
    torch.cuda.synchronize()
    q = env.qpos
    assert torch.isfinite(q).all() and torch.isfinite(env.qvel).all() and torch.isfinite(env._obs_buf).all()
    assert (q[:, 3:7].norm(dim=1) - 1).abs().max() < 1e-5
    assert obs['feet_pos'].reshape(n, 4, 3)[:, :, 2].min() > -0.15
    f = obs['contact_forces'].reshape(n, 4, 3)
    assert f[:, :, 2].min() > -1e-3 and f[:, :, 2].max() < 5e4
    mu = env._friction.clamp(min=1.0).reshape(n, 1)     # calf link geoms mix their own friction (<= 1) with the floor's
    assert bool((f[:, :, :2].norm(dim=2) <= 1.5 * mu * f[:, :, 2] + 1e-3).all())                    # inside the friction pyramid
    assert bool((~(f[:, :, 2] > 1e-6) | (obs['contact_state'] > 0.5)).all())                       # force only where contact_state says so
    assert obs['kinetic_energy'].max() < 1e4 and not bool(trunc.any())
    assert nterm > 50 and int(info['step_num'].min()) < 250                                      # re-spawned envs restarted their counters
    assert info.get('step_num') is info['step_num'] and set(dict(**info)) == {'time', 'step_num', 'invalid_contacts', 'contacts_dropped'}
    assert int(info['contacts_dropped'].max()) == 0   # mini_cheetah (15 collision geoms, condim 1) never reaches the 12-contact capacity here


@pytest.mark.parametrize('robot,scene,sensors', [('aliengo', 'perlin', False), ('go2', 'flat', False), ('hyqreal1', 'random_boxes', True)])
def test_baseline_configs_3_4_5_at_4096_envs_on_the_benchmarked_kernels(robot, scene, sensors):
    """BASELINE.json configs[2] (aliengo on perlin), configs[3] (go2 flat: one 4096-env shard of the 32768) and configs[4] (hyqreal1
    on random_boxes with IMU + HeightMap) at the benchmark's size, torques and auto-reset convention, on the production kernel
    variants bench.py times (height field + exact primitive pairs; elliptic cones; world boxes + cones): a full residency of
    4096 wavefronts with their LDS state, pending flags and load hints.  Properties the domain offers at that size: finite state,
    unit quaternions, feet and base inside the scene, unilateral contact forces inside their friction cone, contact_state
    consistent with the forces, terminated envs are re-spawned and restart their counters, nothing diverges, and the row capacity
    is (almost) never reached."""
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    from gym_quadruped_amd.sensors import IMU, HeightMap
    n = 4096
    names = tuple(QuadrupedEnv.ALL_OBS) + (IMU.ALL_OBS if sensors else ())
    kw = dict(accel_name='Body_Acc', gyro_name='Body_Gyro', imu_site_name='imu', accel_noise=0.01, gyro_noise=0.01, accel_bias_rate=0.01, gyro_bias_rate=0.01, seed=1)
    env = QuadrupedEnv(robot, scene=scene, state_obs_names=names, num_envs=n, solver='newton', solver_iterations=100, solver_tolerance=1e-8,
                       auto_reset='next_step', seed=77, **(dict(sensors=(IMU,), sensors_kwargs=(kw,)) if sensors else {}))
    env.reset(random=True)
    assert not bool(env.lift_failed.any())
    hm = HeightMap(num_rows=5, num_cols=5, dist_x=0.1, dist_y=0.1, mj_model=env.mjModel, mj_data=env) if sensors else None
    g = torch.Generator(device='cuda:0').manual_seed(0)
    nterm = ndrop = 0
    for k in range(200):
        obs, rew, term, trunc, info = env.step(torch.randn(n, 12, generator=g, device='cuda:0') * 50)
        nterm += int(term.sum()); ndrop += int((info['contacts_dropped'] > 0).sum())
        assert not bool(trunc.any()), f'step {k}: an env diverged'
        if hm is not None and k % 20 == 0:
            heights = hm.update_height_map(env.qpos[:, 0:3], yaw=obs['base_ori_euler_xyz'][:, 2])
    torch.cuda.synchronize()
    q = env.qpos
    assert torch.isfinite(q).all() and torch.isfinite(env.qvel).all() and torch.isfinite(env._obs_buf).all()
    assert (q[:, 3:7].norm(dim=1) - 1).abs().max() < 1e-5
    zmax = 1.2 if scene == 'flat' else 2.5
    assert float(q[:, 2].min()) > -0.2 and float(q[:, 2].max()) < zmax + 1.0
    f = obs['contact_forces'].reshape(n, 4, 3)
    assert torch.isfinite(f).all() and float(f.norm(dim=2).max()) < 1e5
    if scene == 'flat':   # the floor's normal is z: unilateral, inside the cone of the mixed friction coefficient
        assert f[:, :, 2].min() > -1e-3
        mu = env._friction.clamp(min=1.0).reshape(n, 1)
        assert bool((f[:, :, :2].norm(dim=2) <= 1.5 * mu * f[:, :, 2] + 1e-2).all())
    assert bool((~(f.norm(dim=2) > 1e-5) | (obs['contact_state'] > 0.5)).all())
    assert nterm > 50 and int(info['step_num'].min()) < 150 and int(env._episode.max()) > 1
    from helpers import tally_note
    tally_note(f'row capacity at 4096 envs, {robot} {scene}: {ndrop} of {n * 200} env-steps ({100.0 * ndrop / (n * 200):.3f} %) had contacts cut by the 12-contact / 63-row capacity; {nterm} terminations')
    # measured on MI355X (profiles/r04_parity_tallies.txt): aliengo perlin and hyqreal1 boxes < 0.1 %, go2 flat 0.27 % (condim-6 feet: 6 rows + 5 reserved per contact)
    assert ndrop <= 0.01 * n * 200, f'{ndrop} env-steps of {n * 200} lost contacts to the row capacity'
    if hm is not None:
        assert tuple(heights.shape) == (n, 5, 5, 1, 3) and torch.isfinite(heights).all() and float(heights[..., 2].max()) > 0.02
        assert float(obs['imu_acc'].abs().max()) > 1.0


def test_feet_contact_state_with_permuted_legs_order():
    """'contact_state' is always FL FR RL RR (quirk B5) while forces follow legs_order; the accessor must label
    both by leg name.  A robot standing on two diagonal feet tells the orders apart."""
    n = 8
    lo = ('FR', 'FL', 'RR', 'RL')
    env = _env('aliengo', n, legs_order=lo, accessors=True, solver='newton', state_obs_names=('qpos', 'contact_state', 'contact_forces'))
    ref = _env('aliengo', n, accessors=True, solver='newton', state_obs_names=('qpos', 'contact_state', 'contact_forces'))
    key = np.asarray(env.mjModel.key_qpos[0]).copy()
    qp = np.tile(key, (n, 1))
    qp[:, 2] = 0.30
    # fold the FL and RR legs up so that only FR and RL can touch
    for leg in ('FL', 'RR'):
        idx = env.legs_qpos_idx[leg]
        qp[:, idx[1]] += 1.2; qp[:, idx[2]] -= 0.6
    for e_ in (env, ref):
        e_.reset(qpos=qp, qvel=np.zeros((n, 18), np.float32))
        for _ in range(40):
            e_.step(torch.zeros(n, 12, device='cuda:0'))
    torch.cuda.synchronize()
    cs, _, grf = env.feet_contact_state('world', ground_reaction_forces=True)
    cs_r, _, grf_r = ref.feet_contact_state('world', ground_reaction_forces=True)
    touched = 0
    for leg in ('FL', 'FR', 'RL', 'RR'):
        assert torch.equal(cs[leg], cs_r[leg]), leg
        assert torch.allclose(grf[leg], grf_r[leg], atol=1e-4), leg
        assert bool((~(grf[leg][:, 2] > 1e-3) | cs[leg]).all()), leg     # a force on a leg implies that leg's state
        touched += int(cs[leg].sum())
    assert touched > 0 and len({tuple(cs[leg].tolist()) for leg in lo}) > 1, 'the pose must tell the legs apart'


def test_state_dict_round_trip_includes_imu_bias_and_resampling_state():
    from gym_quadruped_amd.sensors import IMU
    n = 32
    kw = dict(accel_name='imu_acc', gyro_name='imu_gyro', imu_site_name='imu', accel_noise=0.01, gyro_noise=0.02,
              accel_bias_rate=0.03, gyro_bias_rate=0.04, seed=5)
    mk = lambda: _env('aliengo', n, state_obs_names=('qpos', 'qvel') + IMU.ALL_OBS, sensors=(IMU,), sensors_kwargs=(kw,), solver='newton',
                      auto_reset='next_step', seed=9, base_vel_command_type='forward+reset',
                      external_disturbances_kwargs={'type': 'reset', 'y': (-10.0, 10.0)})
    a, b = mk(), mk()
    a.reset(random=True); b.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(1)
    seq = [torch.randn(n, 12, generator=g, device='cuda:0') * 30 for _ in range(40)]
    for x in seq[:20]:
        a.step(x)
    ck = a.state_dict()
    assert float(ck['sensor_bias'][0].abs().max()) > 0
    for x in seq[20:]:
        oa, *_ = a.step(x)
    final = {k: v.clone() for k, v in oa.items()}
    b.load_state_dict(ck)
    for x in seq[20:]:
        ob, *_ = b.step(x)
    for k in final:
        assert torch.equal(final[k], ob[k]), k
    assert torch.equal(a.sensors[0].bias_state, b.sensors[0].bias_state) and torch.equal(a._h9, b._h9)
    # reset() zeroes the control set-point the getter reports (reference :334)
    a.reset(random=True)
    assert not bool(a.torque_ctrl_setpoint.any())


@pytest.mark.parametrize('shards', [4, 0])
def test_pipelined_rollout_equals_the_step_loop(shards):
    """QuadrupedEnv.rollout (shards > 0: gq_step_range on one stream per shard of envs; shards = 0: ONE persistent launch in
    which every wavefront plays the whole sequence of its env) must leave exactly the state of the plain step loop -
    auto-resets, command redraws and IMU walks included - and deliver every step's observation rows."""
    from gym_quadruped_amd.sensors import IMU
    n, K = 1024, 60
    kw = dict(accel_name='imu_acc', gyro_name='imu_gyro', imu_site_name='imu', accel_noise=0.01, gyro_noise=0.02, accel_bias_rate=0.03, gyro_bias_rate=0.04, seed=5)
    mk = lambda: _env('aliengo', n, state_obs_names=('qpos', 'qvel', 'contact_forces') + IMU.ALL_OBS, sensors=(IMU,), sensors_kwargs=(kw,), solver='newton',
                      auto_reset='next_step', seed=9, base_vel_command_type='random+reset')
    a, b = mk(), mk()
    a.reset(random=True); b.reset(random=True)
    a._h9[:, 1] = 25; b._h9[:, 1] = 25          # command redraws inside the rollout
    g = torch.Generator(device='cuda:0').manual_seed(1)
    acts = torch.randn(K, n, 12, generator=g, device='cuda:0') * 40
    rows = []
    for k in range(K):
        o, _, term, _, _ = a.step(acts[k])
        rows.append(a._obs_buf.clone())
    out = torch.zeros(K, n, b._obs_dim, device='cuda:0')
    b.rollout(acts, shards=shards, obs_out=out)
    torch.cuda.synchronize()
    for k in ('_qpos', '_qvel', '_warm', '_time', '_step_num', '_episode', '_cmd', '_h9', '_terminated', '_obs_buf'):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(a.sensors[0].bias_state, b.sensors[0].bias_state)
    assert torch.equal(torch.stack(rows), out)
    assert int(a._episode.max()) > 1, 'the rollout must contain auto-resets'
    with pytest.raises(ValueError):  # an empty sequence is refused before any launch
        b.rollout(acts[:0], shards=shards)
    # a checkpoint written before the resampling counters moved into one [N, 6] tensor still loads (legacy keys -> columns)
    sd = a.state_dict()
    h9 = sd.pop('_h9')
    sd.update({'_steps_after_vel': h9[:, 0].clone(), '_steps_before_vel': h9[:, 1].clone(), '_steps_after_dist': h9[:, 3].clone(), '_steps_before_dist': h9[:, 4].clone()})
    b._h9[:, [0, 1, 3, 4]] = -7
    b.load_state_dict(sd)
    assert torch.equal(b._h9[:, [0, 1, 3, 4]], h9[:, [0, 1, 3, 4]])


@pytest.mark.parametrize('robot,scene', [('go2', 'flat'), ('hyqreal1', 'random_boxes'), ('aliengo', 'perlin'), ('b2', 'slippery')])
def test_persistent_rollout_equals_the_step_loop_on_the_other_kernel_variants(robot, scene):
    """The persistent rollout (one launch, every wavefront plays all steps of its env) is a kernel variant of its own for every
    scene / cone / geometry combination: elliptic cones (go2), world boxes without and with primitive link geoms (hyqreal1, b2)
    and the height field (aliengo perlin) must also end bit-identical to the step loop, in-kernel re-spawns inside the scene
    included."""
    n, K = 512, 50
    mk = lambda: _env(robot, n, scene=scene, state_obs_names=('qpos', 'qvel', 'contact_forces'), solver='newton', auto_reset='next_step', seed=21)
    a, b = mk(), mk()
    a.reset(random=True); b.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(3)
    acts = torch.randn(K, n, 12, generator=g, device='cuda:0') * 40
    rows = []
    for k in range(K):
        a.step(acts[k])
        rows.append(a._obs_buf.clone())
    out = torch.zeros(K, n, b._obs_dim, device='cuda:0')
    b.rollout(acts, shards=0, obs_out=out)
    torch.cuda.synchronize()
    for k in ('_qpos', '_qvel', '_warm', '_time', '_step_num', '_episode', '_cmd', '_terminated', '_lift_failed', '_obs_buf'):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(torch.stack(rows), out)
    assert torch.isfinite(a.qpos).all()
    assert robot == 'b2' or int(a._episode.max()) > 1, 'the rollout must contain auto-resets'   # (b2 does not fall within 50 steps)


@pytest.mark.parametrize('robot,scene', [('hyqreal1', 'random_boxes'), ('aliengo', 'random_boxes'), ('aliengo', 'perlin')])
def test_box_scene_lift_loop_follows_the_reference_rule_and_how_often_it_leaves_it(robot, scene):
    """QuadrupedEnv.reset on a scene with world geoms beyond the floor: the reference lifts by 1.1 max|dist| per mj_step1 until
    no calf-body contact is left (quadruped_env.py:376-388).  The kernel follows that rule for GQ_LIFT_RULE_ITERS = 4
    iterations and then clears the top of whatever is still touched (the rule converges geometrically out of a
    steep side face and the reference raises after 100 iterations).  Here: the spawn state of every env is rebuilt from the
    Philox draw table, the reference rule is run on the oracle, and (i) every env the rule settles within 4 iterations must
    come out of gq_reset exactly where the oracle's rule + mj_step put it, (ii) the share of envs where the kernel leaves
    the rule is counted, bounded and written to the parity tallies."""
    from oracle.oracle import Oracle
    from scipy.spatial.transform import Rotation
    n, seed = 384, 20250929
    env = _env(robot, n, seed=seed, scene=scene, solver='newton', state_obs_names=('qpos', 'qvel'))
    lim = env.terrain_limits
    env.reset(random=True)
    torch.cuda.synchronize()
    qp, qv = env.qpos.cpu().numpy(), env.qvel.cpu().numpy()
    failed = env.lift_failed.cpu().numpy().astype(bool)
    key = np.asarray(env.mjModel.key_qpos[0])
    hip = float(env.robot_cfg.hip_height)
    o = Oracle(env._mm)
    amp = np.float32(20 * np.pi / 180)

    def calf_contacts():
        o.forward(np.zeros(12), stage=1)
        if not o.ncon:
            return np.zeros(0)
        b = o.get('contact_body').astype(int)
        w = o.get('contact_geom1') < 0      # contacts with world geoms
        return o.get('contact_dist')[w & ((b - 2) % 3 == 2)]
    within = beyond = lifted = same = 0
    for e in range(n):
        u = draws(seed, e, 0)
        q0 = key.copy()
        q0[7:] = key[7:] + (2 * u[0:12] - 1) * amp
        v0 = np.zeros(18); v0[6:] = (2 * u[12:24] - 1) * 0.5
        x = lim[0] + (lim[1] - lim[0]) * float(u[24]); y = lim[2] + (lim[3] - lim[2]) * float(u[25])
        roll, pitch = (2 * u[26] - 1) * np.float32(10 * np.pi / 180), (2 * u[27] - 1) * np.float32(10 * np.pi / 180)
        q0[0], q0[1] = x, y
        q0[3:7] = Rotation.from_euler('xyz', [roll, pitch, np.arctan2(-y, -x)]).as_quat(scalar_first=True)
        z, it = hip, 0
        while it < 100:
            o.set_state(np.r_[q0[:2], z, q0[3:]], v0, np.zeros(18), np.zeros(18), 0.0, -1.0)
            d = calf_contacts()
            if not len(d):
                break
            z += 1.1 * np.abs(d).max()
            it += 1
        lifted += it > 0
        if it > 4:
            beyond += 1      # the kernel switched to "clear the top": higher than the rule, never in contact
            assert failed[e] or qp[e, 2] >= z - 0.05
            continue
        within += 1
        o.set_state(np.r_[q0[:2], z, q0[3:]], v0, np.zeros(18), np.zeros(18), 0.0, -1.0)
        o.step(np.zeros(12))
        ok = np.abs(qp[e] - o.qpos).max() < 5e-6 + 1e-9 * np.abs(o.qpos).max() and np.abs(qv[e] - o.qvel).max() < 5e-4
        same += ok
        assert ok or it >= 3, (e, it, qp[e, :3], o.qpos[:3])   # the 4th iteration's fp32 / fp64 touch test may differ by a hair
    msg = (f'reset lift loop {robot} {scene}: {n} envs, {lifted} lifted, {within} settled by the reference rule within 4 iterations '
           f'({same} bit-for-tolerance equal to the oracle), {beyond} where the kernel leaves the rule, {int(failed.sum())} flagged lift_failed')
    from helpers import tally_note
    tally_note(msg)
    assert lifted >= n // 10 and same >= 0.97 * within and beyond <= 0.1 * n
