"""Env-batch sharding across the GPUs of one node: contiguous, independent shards, no collective.

GPU ``g`` of ``world`` owns global env ids ``[g * envs_per_gpu, (g + 1) * envs_per_gpu)``.  The reset RNG is keyed by
(seed, GLOBAL env id, episode), so a rollout does not depend on how many GPUs the batch is spread over.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    envs_per_gpu: int

    @property
    def env_offset(self) -> int:
        return self.rank * self.envs_per_gpu

    @property
    def global_envs(self) -> int:
        return self.world * self.envs_per_gpu

    def global_ids(self):
        return range(self.env_offset, self.env_offset + self.envs_per_gpu)


def shard_plan(rank: int, world: int, envs_per_gpu: int) -> Shard:
    if not (0 <= rank < world) or envs_per_gpu <= 0:
        raise ValueError(f'bad shard request rank={rank} world={world} envs_per_gpu={envs_per_gpu}')
    return Shard(rank, world, envs_per_gpu)


def aggregate_throughput(per_rank_seconds, steps: int, envs_per_gpu: int) -> float:
    """Whole-job env-steps/s: all ranks' units over the SLOWEST rank's time (bench.py contract)."""
    t = max(per_rank_seconds)
    return len(per_rank_seconds) * envs_per_gpu * steps / t
