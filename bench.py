"""bench.py - headline benchmark: batched QuadrupedEnv.step() throughput (env-steps/sec).

Workload (BASELINE.json configs[1]): mini_cheetah on the flat scene, 4096 envs per GPU, random-action rollout
(50 * N(0,1) torques, what the reference's ``action_space.sample() * 50`` produces), ALL_OBS observations (227
scalars), auto-reset on termination (gymnasium NEXT_STEP convention by default: a terminated env spends its next step
slot on reset() and its mj_step; --auto-reset same_step re-spawns inside the terminating launch), sim_dt = 0.002,
MuJoCo's default Newton solver with its default iteration cap / tolerance (100 / 1e-8); --solver pgs selects PGS.  One "step" = one ``env.step(action)`` over the whole batch, exactly what the reference's ``step``
does for one env: mj_step + observation/termination assembly.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Every rank owns an independent shard of 4096 envs (weak scaling, no data-path collective; the process group only
provides the start/stop barrier and the max-over-ranks time).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
EVENT_STRIDE = 8
DEVICE_WARMUP_STEPS = 300   # steps of a scratch env of the same shape before the measured env is touched (clocks, caches)
STEADY_STEPS = 2000         # a timed window shorter than this is followed by a steady-state window of this many steps


def algorithmic_bytes_per_env_step(obs_dim: int) -> int:
    """HBM bytes one env-step must move (DESIGN.md "Roofline accounting"): state rows in, state + obs rows out."""
    reads = 19 * 8 + 18 * 4 + 12 * 4 + 18 * 4 + 18 * 4 + 4 + 4 + 16 + 4      # qpos f64, qvel, ctrl, warm, applied, time, friction, cmd, step_num
    writes = 19 * 8 + 18 * 4 + 18 * 4 + 18 * 4 + 4 + 4 * obs_dim + 4 + 3 + 4  # qpos, qvel, qacc, warm, time, obs, reward, 3 flags, step_num
    return reads + writes


def pmc_traffic():
    """HBM bytes per step-kernel launch from the last committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (profiles/latest_traffic.json, gfx950 FETCH_SIZE x2 correction applied); counters cannot be read from inside
    this process, so the figure comes from the separate profiling passes of the same command.  Returns (bytes, provenance):
    the provenance names the kernel source the counters were taken on (sha256/16 of the csrc/ sources) and says whether the
    library being benchmarked was built from the same sources - a stale figure is labelled, not silently replayed."""
    p = ROOT / 'profiles' / 'latest_traffic.json'
    try:
        d = json.loads(p.read_text())
        now = kernel_source_hash()
        return d['traffic_bytes_per_launch'], {'profile': d.get('profile', 'profiles/latest_traffic.json'), 'kernel_src_sha16': d.get('kernel_src_sha16'),
                                               'current_kernel_src_sha16': now, 'stale': d.get('kernel_src_sha16') != now}
    except Exception:
        return None, None


def valu_issue(kernel_ms: float, waves_per_simd: int = 4):
    """VALU issue fraction of the headline launch ("report achieved HBM fraction and VALU utilisation"):
    VALU instructions per wave from the last committed rocprofv3 --pmc SQ_INSTS_VALU pass (profiles/latest_sq.json) x the
    waves resident per SIMD x the issue cycles of a wave64 VALU instruction (2 on the SIMD-32 datapath for fp32 - MI355X_MICROARCH.md; 4
    on CDNA3's SIMD-16), over the launch's cycles at the 2.4 GHz peak clock.  Same source-hash staleness check as `traffic`."""
    p = ROOT / 'profiles' / 'latest_sq.json'
    try:
        d = json.loads(p.read_text())
        now = kernel_source_hash()
        per_wave = d['valu_per_wave']
        cyc = per_wave * waves_per_simd * 2
        return {'valu_insts_per_wave': per_wave, 'waves_per_simd': waves_per_simd, 'issue_cycles_per_inst': 2, 'clock_ghz': 2.4,
                'busy_us': cyc / 2.4e3, 'frac': cyc / 2.4e3 / (kernel_ms * 1e3),
                'active_lanes_per_valu': d.get('active_lanes_per_valu'),
                'source': {'profile': d.get('profile'), 'kernel_src_sha16': d.get('kernel_src_sha16'), 'current_kernel_src_sha16': now,
                           'stale': d.get('kernel_src_sha16') != now}}
    except Exception:
        return None


def launch_tail():
    """How much of the headline launch is its tail: (launch end - median wave end) / launch end from the last committed wave-timeline pass
    (tools/wave_timeline.py -> profiles/latest_tail.json; instrumented kernel, one launch) - the number that separates the step loop from the
    persistent rollouts, which have no launch boundary.  Same source-hash staleness check as `traffic`."""
    p = ROOT / 'profiles' / 'latest_tail.json'
    try:
        d = json.loads(p.read_text())
        now = kernel_source_hash()
        return {'tail': d['tail'], 'launch_us': d.get('launch_us'), 'median_wave_end_us': d.get('median_wave_end_us'), 'p99_wave_end_us': d.get('p99_wave_end_us'),
                'source': {'profile': d.get('profile'), 'kernel_src_sha16': d.get('kernel_src_sha16'), 'current_kernel_src_sha16': now, 'stale': d.get('kernel_src_sha16') != now}}
    except Exception:
        return None


def kernel_source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    # the DEVICE side of the step kernel (and the flags it is built with): the host API / model lowering do not move its traffic
    for name in ('Makefile', 'gq_boxes.h', 'gq_convex.h', 'gq_device.h', 'gq_exchange.h', 'gq_heightmap.h', 'gq_kernels.hip', 'gq_model_dev.h', 'gq_newton.h', 'gq_pairs.h', 'gq_step_body.h', 'gq_step_kernel.h'):
        h.update((ROOT / 'gym_quadruped_amd' / 'csrc' / name).read_bytes())
    return h.hexdigest()[:16]


def cpu_baseline(seconds_budget: float = 15.0, all_cores: bool = True):
    """Single-env CPU fp64 restatement of the reference step (oracle, MuJoCo's default Newton solver), 1 core."""
    from gym_quadruped_amd.cabi import ALL_OBS
    from oracle.oracle import Oracle
    from tests.helpers import marshalled

    mm = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-8)
    o = Oracle(mm)
    q = mm.md.key_qpos[0].copy()
    q[2] = 0.3
    o.set_state(q, np.zeros(18), np.zeros(18), np.zeros(18), 0.0, -1.0)
    rng = np.random.default_rng(0)
    chunk, done, t_used = 20000, 0, 0.0
    while t_used < seconds_budget:
        ctrl = rng.normal(0, 1, (chunk, 12)).astype(np.float32).astype(np.float64) * 50
        t0 = time.perf_counter()
        o.rollout(ctrl, ALL_OBS)
        t_used += time.perf_counter() - t0
        done += chunk
    out = {'value': done / t_used, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'{done} steps of 1 mini_cheetah env on flat, 50*N(0,1) torques, ALL_OBS assembled each step, '
                     f'reset to the start state on termination; C fp64 restatement of mj_step with the Newton solver '
                     f'(MuJoCo itself is not installable here; gcc -O3 -march=x86-64-v2, see oracle/Makefile - portable across the build and the GPU box, '
                     f'so conservative for this CPU next to -march=native), {os.cpu_count()} host cores present, 1 used'}
    if all_cores:
        try:
            out['all_cores'] = cpu_baseline_all_cores()
        except Exception as e:  # noqa: BLE001 - a reported extra, never fatal
            out['all_cores'] = {'error': f'{type(e).__name__}: {e}'}
    return out


def _cpu_worker(args):
    """One host process = one oracle env stepping for `budget` seconds (N envs over all host cores)."""
    seed, budget = args
    from gym_quadruped_amd.cabi import ALL_OBS
    from oracle.oracle import Oracle
    from tests.helpers import marshalled
    mm = marshalled('mini_cheetah', solver=1, iterations=100, tolerance=1e-8)
    o = Oracle(mm)
    q = mm.md.key_qpos[0].copy()
    q[2] = 0.3
    o.set_state(q, np.zeros(18), np.zeros(18), np.zeros(18), 0.0, -1.0)
    rng = np.random.default_rng(seed)
    chunk, done = 5000, 0
    ctrl = rng.normal(0, 1, (chunk, 12)).astype(np.float32).astype(np.float64) * 50
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        o.rollout(ctrl, ALL_OBS)
        done += chunk
    return done, time.perf_counter() - t0


def cpu_baseline_all_cores(seconds_budget: float = 6.0):
    """The same CPU restatement with one independent env per host core (no shared state: embarrassingly parallel), all
    cores busy for `seconds_budget`; value = total env-steps / the slowest worker's time."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    with mp.get_context('fork').Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(k, seconds_budget) for k in range(cores)])
    total, tmax = sum(r[0] for r in res), max(r[1] for r in res)
    return {'value': total / tmax, 'unit': 'env-steps/s', 'cores': cores,
            'sample': f'{cores} processes x 1 env each, {total} env-steps in {tmax:.1f} s (slowest worker), same workload as the 1-core figure'}


def config_line(QuadrupedEnv, robot, scene, n, device, pool, imu=False, heightmap=False, steps=400, warmup=100):
    """One BASELINE.json config timed like the headline (untimed warm-up, K steps between two synchronisations, HIP events
    around every EVENT_STRIDE-th step-kernel launch), reported as a secondary of the default run: configs[2..4]."""
    obs_names = tuple(QuadrupedEnv.ALL_OBS)
    sensors = sensors_kwargs = None
    if imu:
        from gym_quadruped_amd.sensors import IMU
        names = {'hyqreal1': ('Body_Acc', 'Body_Gyro')}.get(robot, ('imu_acc', 'imu_gyro'))
        sensors, sensors_kwargs = (IMU,), (dict(accel_name=names[0], gyro_name=names[1], imu_site_name='imu', accel_noise=0.01, gyro_noise=0.01,
                                               accel_bias_rate=0.01, gyro_bias_rate=0.01, seed=1),)
        obs_names = obs_names + IMU.ALL_OBS
    t_build = time.perf_counter()
    env = QuadrupedEnv(robot, state_obs_names=obs_names, scene=scene, num_envs=n, device=device, sensors=sensors, sensors_kwargs=sensors_kwargs,
                       auto_reset='next_step', solver='newton', solver_iterations=100, solver_tolerance=1e-8, seed=1000)
    env.reset(random=True)
    hm = None
    if heightmap:
        from gym_quadruped_amd.sensors import HeightMap
        # the map follows the base (what examples/aliengo_with_heightmap.py does by hand after every step): the step kernel casts the rays
        # (gq_batch_set_heightmap); GQ_BENCH_HM_SEPARATE=1 measures the separate ray kernel behind every step instead
        hm_follow = scene != 'flat' and os.environ.get('GQ_BENCH_HM_SEPARATE', '0') != '1'
        hm = HeightMap(num_rows=5, num_cols=5, dist_x=0.1, dist_y=0.1, mj_model=env.mjModel, mj_data=env, follow_base=hm_follow)
        yaw0 = torch.zeros(n, device=device)

    def one(i, ev=None):
        env._profile_events = ev
        o = env.step(pool[i % 64])[0]
        if hm is not None:
            if hm.follow_base:
                hm.update_height_map()
            else:
                hm.update_height_map(env.qpos[:, 0:3], yaw=o['base_ori_euler_xyz'][:, 2] if 'base_ori_euler_xyz' in o else yaw0)
    for i in range(warmup):
        one(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % EVENT_STRIDE == 0 else None for i in range(steps)]
    torch.cuda.synchronize(device)
    t_build = time.perf_counter() - t_build
    t0 = time.perf_counter()
    for i in range(steps):
        one(i, ev[i])
    env._profile_events = None
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean([p[0].elapsed_time(p[1]) for p in ev if p is not None]))
    b = algorithmic_bytes_per_env_step(env._obs_dim)
    out = {'value': n * steps / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'warmup': warmup, 'kernel_ms': kernel_ms,
           'obs_dim': env._obs_dim, 'bytes_per_env_step': b, 'achieved_gbs': n * b / (kernel_ms * 1e-3) / 1e9,
           'frac': n * b / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           'workload': f'{robot} {scene}, {n} envs, ALL_OBS' + (' + IMU (6 observables)' if imu else '') + ((' + 5x5 HeightMap that follows the base, its rays cast by the step kernel every step' if hm.follow_base else ' + 5x5 HeightMap every step (its ray kernel is inside ms_per_step, not kernel_ms)') if heightmap else '')
                       + ', newton <=100 it tol 1e-8, self-collision on, auto-reset next_step',
           'state_finite': bool(torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all()), 'setup_s': t_build}
    env.close()
    return out


def config_lines(QuadrupedEnv, n, device, pool):
    """BASELINE.json configs[2..4] on this one GPU (config 4 = one 4096-env shard of its 8)."""
    out = {}
    for key, kw in (('cfg3_aliengo_perlin', dict(robot='aliengo', scene='perlin')),
                    ('cfg4_go2_flat_one_shard', dict(robot='go2', scene='flat')),
                    ('cfg5_hyqreal1_boxes_imu_heightmap', dict(robot='hyqreal1', scene='random_boxes', imu=True, heightmap=True))):
        try:
            out[key] = config_line(QuadrupedEnv, n=n, device=device, pool=pool, **kw)
        except Exception as e:  # noqa: BLE001 - a reported extra: the headline line must still be printed
            out[key] = {'error': f'{type(e).__name__}: {e}'}
    return out


def secondary_lines(QuadrupedEnv, n, device, pool, args, steps=400, warmup=100):
    """Short runs of the same workload with (i) the reference's default observation set (73 scalars, SURVEY.md 8d
    "secondary") and (ii) the PGS solver the north-star names; same timing discipline, reported next to the headline."""
    out = {}
    for key, obs_names, solver, sc in (('default_obs', QuadrupedEnv._DEFAULT_OBS, args.solver, _self_collision(args)), ('pgs', tuple(QuadrupedEnv.ALL_OBS), 'pgs', None),
                                       ('pair_exchange_off', tuple(QuadrupedEnv.ALL_OBS), args.solver, None),
                                       ('self_collision_capsule_proxies', tuple(QuadrupedEnv.ALL_OBS), args.solver, 'capsule'),
                                       ('self_collision_off', tuple(QuadrupedEnv.ALL_OBS), args.solver, False)):
        env = QuadrupedEnv('mini_cheetah', state_obs_names=obs_names, scene='flat', num_envs=n, device=device, auto_reset='next_step',
                           solver=solver, solver_iterations=100, solver_tolerance=1e-8, seed=1000, self_collision=sc,
                           pair_exchange=(key != 'pair_exchange_off' and not args.no_pair_exchange))
        env.reset(random=True)
        for i in range(warmup):
            env.step(pool[i % 64])
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            env.step(pool[i % 64])
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        out[key] = {'value': n * steps / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps,
                    'obs_dim': env._obs_dim, 'solver': solver, 'bytes_per_env_step': algorithmic_bytes_per_env_step(env._obs_dim),
                    'self_collision': env._mm.self_collision}
        if key == 'pair_exchange_off':
            out[key]['note'] = ('the headline workload with every env computing its own convex pairs (QuadrupedEnv(pair_exchange=False)): the launch then lasts as long as '
                                'its most entangled robot - csrc/gq_exchange.h; results are bit-identical')
        if key == 'self_collision_capsule_proxies':
            out[key]['note'] = ('the headline workload with the capsule proxies of rounds 2 - 5 in place of the convex routine for robot-robot pairs that involve a mesh '
                                '(QuadrupedEnv(self_collision="capsule")): an approximation - such a contact is found late, by the gap between hull and capsule')
        env.close()
    # the same workload as an OPEN-LOOP rollout (QuadrupedEnv.rollout / gq_rollout): 2 shards of envs on 2 HIP streams, no
    # cross-env barrier between steps - what a random-action / dataset-recording rollout can use and a policy loop cannot.
    # Measured on MI355X: 1 shard 63.2, 2 shards 67.5, 4 shards 36.4, 8 shards 26.8 M env-steps/s (tools/rollout_trace.py):
    # overlapping launches share the SIMDs, so each still lasts as long as its slowest wave under full contention, and more
    # than two queues serialise
    env = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), scene='flat', num_envs=n, device=device, auto_reset='next_step',
                       solver=args.solver, solver_iterations=100, solver_tolerance=1e-8, seed=1000, self_collision=_self_collision(args), pair_exchange=not args.no_pair_exchange)
    env.reset(random=True)
    acts = torch.stack(pool)                       # [64, n, 12]
    for _ in range(4):
        env.rollout(acts, shards=2)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        env.rollout(acts, shards=2)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    # ... and as ONE persistent launch per 64-step sequence (shards=0): every wavefront plays its env's steps back to back
    for _ in range(4):
        env.rollout(acts, shards=0)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    for _ in range(reps):
        env.rollout(acts, shards=0)
    torch.cuda.synchronize(device)
    dtp = time.perf_counter() - t1
    out['persistent_rollout'] = {'value': n * reps * 64 / dtp, 'unit': 'env-steps/s', 'ms_per_step': dtp / (reps * 64) * 1e3, 'steps': reps * 64,
                                 'note': 'open-loop: one launch per 64-step action sequence, each wavefront steps its env 64 times without a launch boundary (gq_rollout shards=0); same results as the step loop, bit for bit'}
    # CLOSED-LOOP persistent rollouts (gq_rollout_closed): a policy in the loop, no launch boundary.  The policy is the library's joint-space
    # PD law; with Gaussian torque noise of the benchmark's amplitude (sigma 50 N m - an exploring policy) the robots fall and re-spawn as
    # under the benchmark's random actions, so the figures compare with the headline; without noise the robots stand on four feet - a
    # different, heavier contact workload (see DESIGN.md section 3 for the like-for-like comparison with the same actions played open loop)
    K = 512
    # gq_rollout_closed evaluates the policy inside the Newton step kernels only: with --solver pgs the block is skipped (GQ_EINVAL otherwise)
    for key, kw in () if args.solver != 'newton' else (('closed_loop_inline', dict(mode='inline', noise_sigma=50.0)), ('closed_loop_mailbox', dict(mode='mailbox', noise_sigma=50.0)),
                    ('closed_loop_inline_standing', dict(mode='inline', noise_sigma=0.0))):
        env.rollout_closed_loop(256, 25.0, 0.8, **kw)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        env.rollout_closed_loop(K, 25.0, 0.8, **kw)
        torch.cuda.synchronize(device)
        dtc = time.perf_counter() - t2
        out[key] = {'value': n * K / dtc, 'unit': 'env-steps/s', 'ms_per_step': dtc / K * 1e3, 'steps': K, 'policy': 'joint-space PD kp 25 kd 0.8 towards keyframe 0' +
                    (', + N(0, 50) torque noise' if kw['noise_sigma'] else ' (robots stand: ~4 feet in contact every step)'),
                    'mean_feet_in_contact': float(env._obs_views['contact_state'].sum(1).mean()),
                    'note': {'inline': 'the stepping wavefront evaluates the policy on the observation row it has just written',
                             'mailbox': 'policy kernel on a second stream; actions / observations through per-env mailboxes, per-XCD ready queues; env-steps are tasks popped by the wavefronts of one persistent launch'}[kw['mode']]
                            + '; states equal the step loop fed with the same actions bit for bit (tests/test_gpu_closed_loop.py)'}
    # the regime a controller / RL user lives in, as a STEP LOOP: the caller computes joint-space PD torques towards keyframe 0 from the observation
    # it has just received (torch, on the GPU) and plays them through env.step - robots stand on four feet; compare closed_loop_inline_standing
    if args.solver == 'newton':
        env.reset(random=True)
        qdes = torch.as_tensor(np.asarray(env.mjModel.key_qpos[0][7:], dtype=np.float32), device=device)
        o = env.step(torch.zeros(n, 12, device=device))[0]
        for _ in range(200):
            o = env.step(25.0 * (qdes - o['qpos_js']) - 0.8 * o['qvel_js'])[0]
        torch.cuda.synchronize(device)
        t3 = time.perf_counter()
        for _ in range(steps):
            o = env.step(25.0 * (qdes - o['qpos_js']) - 0.8 * o['qvel_js'])[0]
        torch.cuda.synchronize(device)
        dts = time.perf_counter() - t3
        out['step_loop_standing'] = {'value': n * steps / dts, 'unit': 'env-steps/s', 'ms_per_step': dts / steps * 1e3, 'steps': steps,
                                     'mean_feet_in_contact': float(env._obs_views['contact_state'].sum(1).mean()),
                                     'policy': 'joint-space PD kp 25 kd 0.8 towards keyframe 0, evaluated by the caller in torch between two env.step calls',
                                     'note': 'QuadrupedEnv.step in a loop with a policy in it (robots stand: ~4 feet in contact every step); the two torch kernels of the law are inside the time'}
    out['pipelined_rollout'] = {'value': n * reps * 64 / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / (reps * 64) * 1e3, 'steps': reps * 64, 'shards': 2,
                                'note': 'open-loop: each shard of 2048 envs chains its steps on its own stream (gq_step_range); same kernels and results as the step loop'}
    env.close()
    # two independent QuadrupedEnv batches on two HIP streams of this process (examples/two_stream_rollout.py): each batch's
    # launch tail runs under the other batch's bulk.  2 x n envs = twice the per-GPU env count of the headline - reported as
    # what one GPU delivers when the env count is free, not as the headline metric
    envs, streams = [], []
    for k in range(2):
        st = torch.cuda.Stream(device=device)
        with torch.cuda.stream(st):
            e = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), scene='flat', num_envs=n, device=device, auto_reset='next_step',
                             solver=args.solver, solver_iterations=100, solver_tolerance=1e-8, seed=1000 + k, env_id_offset=k * n,
                             self_collision=_self_collision(args), pair_exchange=not args.no_pair_exchange)
            e.reset(random=True)
        envs.append(e); streams.append(st)
    torch.cuda.synchronize(device)
    def both(nsteps):
        for i in range(nsteps):
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    envs[k].step(pool[(i + 7 * k) % 64])
        torch.cuda.synchronize(device)
    both(warmup)
    t0 = time.perf_counter()
    both(steps)
    dt = time.perf_counter() - t0
    out['two_batches_two_streams'] = {'value': 2 * n * steps / dt, 'unit': 'env-steps/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'total_envs': 2 * n,
                                      'note': 'two independent batches of envs_per_gpu envs each, one HIP stream each, same process and GPU'}
    for e in envs:
        e.close()
    return out


def _self_collision(args):
    """QuadrupedEnv(self_collision=...) of the command line: None = the default (on with the Newton solver, mesh pairs by the convex routine)."""
    if args.no_self_collision or args.self_collision == 'off':
        return False
    return 'capsule' if args.self_collision == 'capsule' else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--obs', choices=['all', 'default'], default='all')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the short secondary measurements (_DEFAULT_OBS, PGS)')
    ap.add_argument('--auto-reset', choices=['next_step', 'same_step', 'off'], default='next_step')
    ap.add_argument('--no-auto-reset', action='store_true')
    ap.add_argument('--solver', choices=['newton', 'pgs'], default='newton')
    ap.add_argument('--scene', default='flat', help="headline metric: flat; box scenes (random_boxes, stairs, ...) for the secondary configs")
    ap.add_argument('--robot', default='mini_cheetah', help='headline metric: mini_cheetah; other registry robots for the secondary configs')
    ap.add_argument('--no-pair-exchange', action='store_true', help='every env computes its own convex self pairs (default: shared with idle wavefronts of the launch, csrc/gq_exchange.h)')
    ap.add_argument('--no-self-collision', action='store_true', help='switch robot self-collision off (MuJoCo default and default here: on)')
    ap.add_argument('--self-collision', choices=['convex', 'capsule', 'off'], default='convex',
                    help="robot-robot pairs with a mesh / cylinder: 'convex' = MuJoCo's general convex routine on the hulls (default, the headline), "
                         "'capsule' = capsule proxies in their place (approximation), 'off' = no robot-robot contacts")
    ap.add_argument('--imu', action='store_true', help='BASELINE config 5: IMU plug-in (6 observables; robots that expose accelerometer + gyro sensors)')
    ap.add_argument('--heightmap', action='store_true', help='BASELINE config 5: a 5x5 HeightMap @ 0.1 m updated every step')
    ap.add_argument('--dist-backend', default='nccl', help=argparse.SUPPRESS)   # test hook: 'gloo' together with GQ_BENCH_SHARE_DEVICE=1 runs the
    args = ap.parse_args()                                                       # N-rank protocol with every rank on GPU 0 (one-GPU boxes)

    from gym_quadruped_amd.sharding import aggregate_throughput, shard_plan
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    shard = shard_plan(rank, world, args.envs_per_gpu)   # contiguous, independent shards; no data-path collective
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} != WORLD_SIZE {world}')
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...')
    # the CPU baseline runs first: its all-cores leg forks worker processes, which must happen before this process
    # creates a HIP context
    # N > 1: rank 0 still times the 1-core leg (the other ranks wait for it at the rendezvous; nothing of it is inside the
    # timed region), so that a scaling line carries its CPU baseline too; the all-cores leg stays with the N = 1 line
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline() if world == 1 else cpu_baseline(seconds_budget=10.0, all_cores=False)
    dist = None
    # GQ_BENCH_FORCE_DIST=1 (tests/test_gpu_boundary.py): a ONE-rank torchrun launch also goes through the process-group path - the RCCL
    # initialisation, the barriers around the timed region and the MAX all-reduce of the N > 1 protocol - on a one-GPU box
    if world > 1 or os.environ.get('GQ_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if os.environ.get('GQ_BENCH_SHARE_DEVICE') == '1':
            local_rank = 0   # protocol test: all ranks on one GPU (the throughput it prints is NOT a scaling number)
        if torch.cuda.device_count() < world and os.environ.get('GQ_BENCH_SHARE_DEVICE') != '1':
            raise SystemExit(f'--gpus {world} needs {world} visible GPUs, this node shows {torch.cuda.device_count()} (GQ_BENCH_SHARE_DEVICE=1 runs the '
                             f'N-rank protocol on one GPU: a protocol test, not a scaling number)')
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device(f'cuda:{local_rank}'))
        else:
            dist.init_process_group(backend=args.dist_backend)
    device = torch.device(f'cuda:{local_rank}')
    torch.cuda.set_device(device)

    from gym_quadruped_amd.quadruped_env import QuadrupedEnv

    obs_names = tuple(QuadrupedEnv.ALL_OBS) if args.obs == 'all' else QuadrupedEnv._DEFAULT_OBS
    n = args.envs_per_gpu
    sensors, sensors_kwargs = None, None
    if args.imu:   # examples/aliengo_with_imu.py:23-31 (noise 0.01, bias rate 0.01); sensor names per robot
        from gym_quadruped_amd.sensors import IMU
        names = {'hyqreal1': ('Body_Acc', 'Body_Gyro')}.get(args.robot, ('imu_acc', 'imu_gyro'))
        sensors, sensors_kwargs = (IMU,), (dict(accel_name=names[0], gyro_name=names[1], imu_site_name='imu', accel_noise=0.01, gyro_noise=0.01,
                                               accel_bias_rate=0.01, gyro_bias_rate=0.01, seed=1),)
        obs_names = obs_names + IMU.ALL_OBS
    env = QuadrupedEnv(args.robot, state_obs_names=obs_names, scene=args.scene, num_envs=n, device=device, sensors=sensors, sensors_kwargs=sensors_kwargs,
                       auto_reset=False if (args.no_auto_reset or args.auto_reset == 'off') else args.auto_reset, solver=args.solver, solver_iterations=100, solver_tolerance=1e-8,
                       seed=1000, env_id_offset=shard.env_offset, self_collision=_self_collision(args), pair_exchange=not args.no_pair_exchange)  # shards: disjoint global env ids -> disjoint RNG counters
    env.reset(random=True)
    g = torch.Generator(device=device).manual_seed(rank)
    pool = [torch.randn(n, 12, generator=g, device=device) * 50 for _ in range(64)]

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    # device warm-up, not steps of the measured rollout: a GPU that has been idle while the host built the model sits in a
    # low power state, and the first launches of a kernel pay instruction-cache and TLB misses.  A scratch env of the same
    # shape runs 300 steps of the same kernel and is thrown away, so that a --steps 20 --warmup 5 run measures the same
    # clocks and caches as a --steps 2000 one; the measured env then does exactly --warmup untimed steps
    scratch = QuadrupedEnv(args.robot, state_obs_names=obs_names, scene=args.scene, num_envs=n, device=device, sensors=sensors, sensors_kwargs=sensors_kwargs,
                           auto_reset='next_step', solver=args.solver, solver_iterations=100, solver_tolerance=1e-8, seed=999,
                           self_collision=_self_collision(args), pair_exchange=not args.no_pair_exchange)
    scratch.reset(random=True)
    for i in range(DEVICE_WARMUP_STEPS):
        scratch.step(pool[i % 64])
    torch.cuda.synchronize(device)
    scratch.close()
    del scratch
    hm = None
    if args.heightmap:   # examples/aliengo_with_heightmap.py:25
        from gym_quadruped_amd.sensors import HeightMap
        hm_follow = args.scene != 'flat' and os.environ.get('GQ_BENCH_HM_SEPARATE', '0') != '1'   # (see config_line)
        hm = HeightMap(num_rows=5, num_cols=5, dist_x=0.1, dist_y=0.1, mj_model=env.mjModel, mj_data=env, follow_base=hm_follow)
        yaw0 = torch.zeros(n, device=device)

    def update_hm(o_i):
        if hm.follow_base:
            hm.update_height_map()
        else:
            hm.update_height_map(env.qpos[:, 0:3], yaw=o_i['base_ori_euler_xyz'][:, 2] if 'base_ori_euler_xyz' in o_i else yaw0)
    for i in range(args.warmup):
        env.step(pool[i % 64])
    # timed region: exactly K steps bracketed by barrier + synchronize
    # HIP events around the kernel launch of every EVENT_STRIDE-th step (on the launch stream): an event pair per launch
    # puts two extra packets between consecutive kernels and costs ~5 % of the step time it is there to measure
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % EVENT_STRIDE == 0 else None
          for i in range(args.steps)]
    env._profile_events = None
    barrier()
    t0 = time.perf_counter()
    nterm = 0
    for i in range(args.steps):
        env._profile_events = ev[i]  # HIP events around the step-kernel launch on the launch stream
        o_i = env.step(pool[i % 64])[0]
        if hm is not None:
            update_hm(o_i)
    env._profile_events = None
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device if args.dist_backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = float(np.mean([p[0].elapsed_time(p[1]) for p in ev if p is not None]))
    # steady state: a short timed window (the driver's 20 steps) sits at the start of a fresh rollout, before most robots have
    # fallen for the first time; the same env simply keeps stepping for STEADY_STEPS more steps and that rate is reported next
    # to the headline (when the timed window is that long already, it IS the steady-state figure)
    steady = None
    if args.steps < STEADY_STEPS and not args.no_secondary:
        barrier()
        t1 = time.perf_counter()
        for i in range(STEADY_STEPS):
            o_i = env.step(pool[i % 64])[0]
            if hm is not None:
                update_hm(o_i)
        barrier()
        dts = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([dts], device=device if args.dist_backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        steady = {'value': world * n * STEADY_STEPS / dts, 'unit': 'env-steps/s', 'ms_per_step': dts / STEADY_STEPS * 1e3, 'steps': STEADY_STEPS,
                  'note': f'the same envs, steps {args.warmup + args.steps}..{args.warmup + args.steps + STEADY_STEPS} of the rollout'}
    finite = bool(torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all())

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and args.robot == 'mini_cheetah' and args.scene == 'flat':
        secondary = secondary_lines(QuadrupedEnv, n, device, pool, args)
        if args.solver == 'newton' and args.obs == 'all' and n == ENVS_PER_GPU:
            secondary.update(config_lines(QuadrupedEnv, n, device, pool))
    if rank == 0:
        total_envs = shard.global_envs
        value = aggregate_throughput([dt] * world, args.steps, n)   # dt is already the max over ranks
        bytes_step = algorithmic_bytes_per_env_step(env._obs_dim)
        achieved = n * bytes_step / (kernel_ms * 1e-3) / 1e9
        # the committed counter passes were taken on the headline workload: replayed for that workload only
        headline = (args.robot == 'mini_cheetah' and args.scene == 'flat' and args.solver == 'newton' and args.obs == 'all' and n == ENVS_PER_GPU
                    and _self_collision(args) is None and not args.imu and not args.heightmap and args.auto_reset == 'next_step' and not args.no_auto_reset)
        traffic = pmc_traffic() if headline else (None, {'note': 'HBM counters are committed for the headline workload only (profiles/rNN_hbm_counters.md)'})
        out = {
            'metric': 'env-steps/sec (batched)', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'device_warmup_steps': DEVICE_WARMUP_STEPS, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': f'{args.robot} {args.scene}, {n} envs/GPU, random-action rollout (50*N(0,1) torques), '
                                   f'{"ALL_OBS" if args.obs == "all" else "_DEFAULT_OBS"} ({env._obs_dim} scalars), '
                                   f'auto-reset on termination ({env.auto_reset_mode}), sim_dt 0.002, {args.solver} solver <=100 it tol 1e-8, robot self-collision {"off" if _self_collision(args) is False or args.solver == "pgs" else ("on (mesh pairs: capsule proxies)" if _self_collision(args) == "capsule" else "on (mesh pairs: GJK / EPA on the hulls, as mjc_Convex)")}'
                                   + (', IMU plug-in' if args.imu else '') + ((', 5x5 HeightMap following the base (rays cast by the step kernel)' if hm.follow_base else ', 5x5 HeightMap every step') if args.heightmap else ''),
                       'envs_per_gpu': n, 'total_envs': total_envs, 'parallelism': f'env-shards x{world} (no collectives)',
                       'state_finite': finite},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic[0], 'traffic_source': traffic[1], 'kernel': 'gq::step_kernel',
                         'kernel_ms': kernel_ms, 'bytes_per_env_step': bytes_step,
                         'valu_issue': valu_issue(kernel_ms) if headline else None,
                         'tail': launch_tail() if headline else None,
                         'note': 'algorithmic bytes / HIP-event kernel time; the step is latency/VALU bound, not HBM bound'},
        }
        if steady is not None:
            secondary = dict(secondary or {}, steady_state=steady)
        elif args.steps >= STEADY_STEPS:
            secondary = dict(secondary or {}, steady_state={'value': value, 'unit': 'env-steps/s', 'steps': args.steps, 'note': 'the timed window itself'})
        if secondary:
            out['secondary'] = secondary
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
