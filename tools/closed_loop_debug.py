"""Where does a closed-loop rollout leave the step loop?  usage: closed_loop_debug.py robot scene n K mode [trials]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv
robot, scene, n, K, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
trials = int(sys.argv[6]) if len(sys.argv) > 6 else 2
import os
mk = lambda: QuadrupedEnv(robot, scene=scene, num_envs=n, device='cuda:0', solver='newton', auto_reset='next_step', seed=11,
                          state_obs_names=('qpos_js', 'qvel_js', 'tau_ctrl_setpoint', 'base_lin_vel', 'contact_forces'))
tot = 0
for trial in range(trials):
    a, b = mk(), mk()
    a.reset(random=True); b.reset(random=True)
    g = torch.Generator(device='cuda:0').manual_seed(2)
    for _ in range(30):
        act = torch.randn(n, 12, generator=g, device='cuda:0') * 40
        a.step(act); b.step(act)
    r = b.rollout_closed_loop(K, 25.0, 0.8, mode=mode, record_obs=True, record_actions=True)
    acts = r['actions']
    first = {}
    hist = []
    for k in range(K):
        a.step(acts[k])
        hist.append((a._obs_buf[:, 24:36].clone(), a._terminated.clone(), a._episode.clone()))
        bad = (a._obs_buf != r['obs_seq'][k]).any(dim=1).nonzero().flatten().tolist()
        for e in bad:
            if e not in first:
                tau_a, tau_b = a._obs_buf[e, 24:36], r['obs_seq'][k][e, 24:36]
                prev = r['obs_seq'][k - 1][e, 24:36] if k else tau_b
                first[e] = (k, 'ACTION differs' + (' (= the previous step\'s)' if torch.equal(tau_b, prev) else '') if not torch.equal(tau_a, tau_b) else 'same action, STATE differs')
    torch.cuda.synchronize()
    if int(os.environ.get('GQ_MB_FLAGS', '0')) & 32:
        import ctypes as C, numpy as np
        cen = np.zeros(n, np.int32)
        b._L.gq_mailbox_census.argtypes = [C.c_void_p, C.c_void_p]
        b._L.gq_mailbox_census(b._hbatch, cen.ctypes.data)
        nx = np.array([sum(1 for x in range(8) if (int(c) >> (4 * x)) & 15) for c in cen])
        print(f'   XCD census: envs stepped by 1 XCD: {(nx == 1).sum()}, by more: {(nx > 1).sum()}, by none {(nx == 0).sum()}; queue = env % 8 for all: {all(((int(c) >> (4 * (e % 8))) & 15) for e, c in enumerate(cen))}')
    for e, (k, why) in sorted(first.items())[:3]:
        for kk in range(max(0, k - 2), min(K, k + 2)):
            ta, tb = hist[kk][0][e], r['obs_seq'][kk][e, 24:36]
            print(f'    env {e} step {kk}: a term {int(hist[kk][1][e])} epi {int(hist[kk][2][e])} tau_a {ta[:4].tolist()} | tau_b {tb[:4].tolist()} | recorded {acts[kk][e, :4].tolist()}')
    tot += len(first)
    print(f'flags {os.environ.get("GQ_MB_FLAGS", "0")} trial {trial}: {robot} {scene} n={n} K={K} {mode}: {len(first)} envs leave the step loop: {sorted(first.items())[:6]}', flush=True)
print(f'flags {os.environ.get("GQ_MB_FLAGS", "0")} {robot} {scene}: total {tot} of {trials * n * K} env-steps')
