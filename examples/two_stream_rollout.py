"""Two independent env batches on two HIP streams of one process: each batch's launch tail (the few envs that need many
Newton iterations) runs under the other batch's bulk, so one GPU delivers more env-steps/s than with a single batch of
the same total size.  This is the closed-loop-friendly form of `QuadrupedEnv.rollout` / `gq_rollout`: a learner can run
its policy on batch A's observations while batch B steps.

    python examples/two_stream_rollout.py [envs_per_batch] [steps]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gym_quadruped_amd.quadruped_env import QuadrupedEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device('cuda:0')
envs, streams, pools = [], [], []
for k in range(2):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        e = QuadrupedEnv('mini_cheetah', state_obs_names=tuple(QuadrupedEnv.ALL_OBS), num_envs=n, device=dev, auto_reset='next_step',
                         seed=k, env_id_offset=k * n)
        e.reset(random=True)
        g = torch.Generator(device=dev).manual_seed(k)
        pools.append([torch.randn(n, 12, generator=g, device=dev) * 50 for _ in range(32)])
    envs.append(e); streams.append(s)


def run(which, nsteps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(nsteps):
        for k in which:
            with torch.cuda.stream(streams[k]):
                envs[k].step(pools[k][i % 32])
    torch.cuda.synchronize()
    return len(which) * n * nsteps / (time.perf_counter() - t0)


run([0, 1], 200)   # warm-up
one = run([0], steps)
two = run([0, 1], steps)
print(f'one batch of {n} envs: {one / 1e6:.1f} M env-steps/s;  two batches of {n} envs on two streams: {two / 1e6:.1f} M env-steps/s total')
