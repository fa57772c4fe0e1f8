"""GPU tests of the closed-loop persistent rollout (include/gq.h gq_rollout_closed, QuadrupedEnv.rollout_closed_loop): the
control loop `a = policy(obs); obs, ... = env.step(a)` around quadruped_env.py:251-307 (README.md:31-33) played without launch
boundaries - env-steps as tasks, a policy kernel on a second stream, per-XCD ready queues.

* the states, flags and observation rows afterwards equal those of the plain step loop fed with the SAME actions, bit for bit -
  also when there are four times more envs than the device has wavefront slots (an env is not bound to a wavefront);
* the recorded actions are the PD law applied to the recorded observations, bit for bit (the loop really is closed);
* a rollout whose policy never answers ends with an error after the deadline - it does not hang - and the batch stays usable.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STATE = ('_qpos', '_qvel', '_qacc', '_warm', '_time', '_step_num', '_episode', '_cmd', '_terminated', '_truncated', '_invalid', '_obs_buf', '_friction',
         '_contacts_dropped')


def _env(robot='mini_cheetah', n=4096, scene='flat', **kw):
    from gym_quadruped_amd.quadruped_env import QuadrupedEnv
    kw.setdefault('state_obs_names', ('qpos_js', 'qvel_js', 'base_lin_vel', 'contact_forces'))
    return QuadrupedEnv(robot, scene=scene, num_envs=n, device='cuda:0', solver='newton', auto_reset='next_step', **kw)


def _twin(robot, n, scene='flat', warm=30, **kw):
    a, b = _env(robot, n, scene, seed=11, **kw), _env(robot, n, scene, seed=11, **kw)
    a.reset(random=True); b.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(2)
    for _ in range(warm):   # some envs are on the ground, some about to re-spawn
        act = torch.randn(n, 12, generator=g, device='cuda:0') * 40
        a.step(act); b.step(act)
    torch.cuda.synchronize()
    for k in STATE:
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    return a, b


@pytest.mark.parametrize('mode', ['mailbox', 'inline'])
@pytest.mark.parametrize('robot,scene,n,K', [('mini_cheetah', 'flat', 4096, 120), ('mini_cheetah', 'flat', 16384, 24), ('go2', 'flat', 1000, 60),
                                              ('aliengo', 'random_boxes', 777, 40)])
def test_closed_loop_rollout_equals_the_step_loop_with_the_same_actions(robot, scene, n, K, mode):
    a, b = _twin(robot, n, scene)
    kp, kd = 25.0, 0.8
    r = b.rollout_closed_loop(K, kp, kd, mode=mode, record_obs=True, record_actions=True)
    code, _, played = b.closed_loop_status()
    assert code == 0 and (mode == 'inline' or played == n * K)
    acts = r['actions']
    rows = []
    for k in range(K):
        a.step(acts[k])
        rows.append(a._obs_buf.clone())
    torch.cuda.synchronize()
    for k in STATE:
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(torch.stack(rows), r['obs_seq'])
    assert torch.isfinite(b.qpos).all()
    assert int(b._episode.max()) > 1, 'the rollout must contain auto-resets'
    # the loop is closed: action k is the PD law on the observation after step k - 1 (k = 0: the row the rollout started from is gone,
    # so the check starts at 1), each operation rounded as in the elementwise expression
    qd = torch.as_tensor(np.asarray(b.mjModel.key_qpos[0][7:19], dtype=np.float32), device='cuda:0')
    o = r['obs_seq']
    q, v = o[:-1, :, 0:12], o[:-1, :, 12:24]
    want = kp * (qd - q) - kd * v
    assert torch.equal(want, acts[1:])   # (an env that re-spawns in step k ignores its action - next-step auto-reset - but it is derived and recorded)
    # and the batch is an ordinary batch afterwards
    a.step(acts[0]); b.step(acts[0])
    torch.cuda.synchronize()
    assert torch.equal(a._qpos, b._qpos) and torch.equal(a._obs_buf, b._obs_buf)


@pytest.mark.parametrize('mode', ['mailbox', 'inline'])
@pytest.mark.parametrize('n', [1, 7, 63, 65])
def test_closed_loop_rollout_on_small_batches_and_single_steps(n, mode):
    """Fewer envs than XCDs (a stepping wavefront pops from ITS XCD's queue only: every XCD needs one, also for one env) and rollouts
    of ONE step (the inline policy is evaluated by the persistent kernel variant only): equal to the step loop, and the first
    recorded action is the PD law on the observation the rollout started from."""
    a, b = _env(n=n, seed=3), _env(n=n, seed=3)
    a.reset(random=True); b.reset(random=True)
    kp, kd = 25.0, 0.8
    qd = torch.as_tensor(np.asarray(b.mjModel.key_qpos[0][7:19], dtype=np.float32), device='cuda:0')
    for K in (1, 5, 1):
        before = b._obs_buf.clone()
        r = b.rollout_closed_loop(K, kp, kd, mode=mode, record_actions=True)
        code, _, played = b.closed_loop_status()
        assert code == 0 and (mode == 'inline' or played == n * K)
        acts = r['actions']
        assert torch.equal(acts[0], kp * (qd - before[:, 0:12]) - kd * before[:, 12:24])
        for k in range(K):
            a.step(acts[k])
        torch.cuda.synchronize()
        for f in STATE:
            assert torch.equal(getattr(a, f), getattr(b, f)), (K, f)


@pytest.mark.parametrize('n,step_waves', [(64, 4096), (9, 2048), (200, 8192)])
def test_mailbox_rollout_with_many_more_step_wavefronts_than_envs(n, step_waves):
    """Pop tickets run ahead of the pushes by the number of resident step wavefronts: with step_waves / n_queues > queue_capacity
    (64 slots for a small batch) tickets t, t + qcap, t + 2 qcap ... wait on ONE slot.  Each must take the item of its own lap
    (lap tag in the item, gq.h) - an env stepped by two wavefronts at once would break the bit equality with the step loop and
    the played count."""
    a, b = _env(n=n, seed=5), _env(n=n, seed=5)
    a.reset(random=True); b.reset(random=True)
    K = 150
    r = b.rollout_closed_loop(K, 25.0, 0.8, mode='mailbox', record_actions=True, step_waves=step_waves, noise_sigma=30.0)
    code, _, played = b.closed_loop_status()
    assert code == 0 and played == n * K
    for k in range(K):
        a.step(r['actions'][k])
    torch.cuda.synchronize()
    for f in STATE:
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert int(b._step_num.max()) <= K


def test_closed_loop_rollout_refuses_ticket_counter_overflow():
    from gym_quadruped_amd import _lib
    env = _env('mini_cheetah', 4096)
    env.reset(random=True)
    with pytest.raises(_lib.GqError, match='32-bit ticket'):
        env.rollout_closed_loop(2 ** 31 - 1, 20.0, 0.5, mode='mailbox')
    env.rollout_closed_loop(3, 20.0, 0.5, mode='mailbox')   # and the batch still works


def test_closed_loop_rollout_with_a_silent_policy_fails_loudly_and_leaves_the_batch_usable():
    """pd = NULL: the caller promises to run the policy side itself.  Nobody does here: every step wavefront waits for a queue item
    that never comes, the deadline (0.3 s) passes, the abort word goes up, the launch ends, and the status call reports it."""
    from gym_quadruped_amd import _lib
    env = _env('mini_cheetah', 4096)
    env.reset(random=True)
    before = env._qpos.clone()
    stream = torch.cuda.current_stream(env.device).cuda_stream
    import time
    t0 = time.perf_counter()
    _lib.check(env._L.gq_rollout_closed(env._hbatch, 10, 0, None, 0, 0, 0.3, env._st, env._out, env._auto_cfg, env._episode.data_ptr(), env._lift_failed.data_ptr(),
                                        None, None, stream), 'gq_rollout_closed')
    with pytest.raises(_lib.GqError, match='aborted'):
        env.closed_loop_status()
    assert time.perf_counter() - t0 < 20.0
    assert torch.equal(before, env._qpos), 'no env was stepped'
    # a proper closed-loop rollout and a plain step still work on the same batch
    env.rollout_closed_loop(5, 20.0, 0.5)
    env.step(torch.zeros(4096, 12, device='cuda:0'))
    torch.cuda.synchronize()
    assert torch.isfinite(env.qpos).all() and int(env._step_num.max()) >= 6


def test_closed_loop_rollout_refuses_what_it_cannot_do():
    from gym_quadruped_amd import _lib
    env = _env('mini_cheetah', 64, state_obs_names=('base_pos',))   # no joint observables in the row: the PD policy has nothing to read
    env.reset(random=True)
    with pytest.raises(_lib.GqError, match='qpos_js'):
        env.rollout_closed_loop(3, 20.0, 0.5)
    env2 = _env('mini_cheetah', 64)
    env2.reset(random=True)
    env2.enable_debug(8)
    with pytest.raises(_lib.GqError, match='production kernel'):
        env2.rollout_closed_loop(3, 20.0, 0.5)


def test_mailbox_view_describes_the_protocol_tables():
    """gq_mailbox_get: what a caller-provided policy kernel needs - the mailbox / queue pointers, the queue geometry and the XCD -> queue
    map (the rule of include/gq.h: the producer of env e's action runs on the XCD of queue e mod n_queues)."""
    from gym_quadruped_amd import _lib
    from gym_quadruped_amd.cabi import GqMailboxView
    env = _env('mini_cheetah', 1000)
    env.reset(random=True)
    v = GqMailboxView()
    _lib.check(env._L.gq_mailbox_get(env._hbatch, C.byref(v)), 'gq_mailbox_get')
    assert v.action and v.steps_done and v.queue_items and v.queue_counters and v.status
    assert 1 <= v.n_queues <= 16 and v.queue_capacity >= 1000 and v.queue_capacity & (v.queue_capacity - 1) == 0 and v.counter_stride == 32
    used = sorted(set(v.xcc_queue[x] for x in range(16)))
    assert used[0] == 0 and used[-1] == v.n_queues - 1
    # the tables are the ones the library's own rollout uses: after one, every env has published its step count
    env.rollout_closed_loop(7, 20.0, 0.5, mode='mailbox')
    torch.cuda.synchronize()
    from gym_quadruped_amd.accessors import _DevPtr
    done = torch.as_tensor(_DevPtr(v.steps_done, (1000,), '<i4'), device='cuda:0')
    assert bool((done == 7).all())
