"""N>1 path on CPU: two gloo ranks run the bench's sharding / timing protocol (barrier, max-over-ranks time,
whole-job aggregate) with the kernel under the emulator standing in for the GPU step."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from helpers import default_reset_cfg, emu_reset, emu_step, marshalled
    from gym_quadruped_amd.sharding import aggregate_throughput, shard_plan
    sh = shard_plan(rank, world, envs_per_gpu=3)
    mm = marshalled('mini_cheetah', iterations=10, terrain_limits=(5, -5, 5, -5))
    # RNG keyed by GLOBAL env id: emulate by offsetting the episode-independent env index through the mask trick
    cfg = default_reset_cfg(seed=42)
    n_glob = sh.global_envs
    mask = np.zeros(n_glob, np.uint8); mask[list(sh.global_ids())] = 1
    st = emu_reset(mm, n_glob, cfg, mask=mask)                 # block index == global env id
    qpos, qvel = st['qpos'][mask == 1].copy(), st['qvel'][mask == 1].copy()
    dist.barrier()
    for _ in range(3):
        out = emu_step(mm, np.zeros((3, 12)), qpos, qvel)
        qpos, qvel = out['qpos'], out['qvel']
    dist.barrier()
    t = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)  # pretend rank 1 was slower
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, (sh.env_offset, qpos[:, :3].tolist()))
    if rank == 0:
        q.put((float(t.item()), gathered, aggregate_throughput([0.5, 0.75], steps=3, envs_per_gpu=3)))
    dist.destroy_process_group()


def test_two_rank_shards_are_disjoint_and_aggregate_uses_slowest_rank():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tmax, gathered, agg = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 0.75 and abs(agg - 2 * 3 * 3 / 0.75) < 1e-9
    assert [g[0] for g in gathered] == [0, 3]
    a, b = np.array(gathered[0][1]), np.array(gathered[1][1])
    assert a.shape == b.shape == (3, 3) and not np.allclose(a, b)     # different global env ids -> different spawns
    # global-id keyed RNG: the same shard recomputed in one process gives the same states
    sys.path.insert(0, str(ROOT / 'tests'))
    from helpers import default_reset_cfg, emu_reset, marshalled
    mm = marshalled('mini_cheetah', iterations=10, terrain_limits=(5, -5, 5, -5))
    st = emu_reset(mm, 6, default_reset_cfg(seed=42))
    assert np.all(np.abs(st['qpos'][3:, :2] - b[:, :2]) < 0.05)        # x,y drift only by the 3 integration steps


def test_shard_plan_validation():
    from gym_quadruped_amd.sharding import shard_plan
    s = shard_plan(3, 8, 4096)
    assert s.env_offset == 12288 and s.global_envs == 32768 and list(s.global_ids())[-1] == 16383
    with pytest.raises(ValueError):
        shard_plan(8, 8, 4096)
