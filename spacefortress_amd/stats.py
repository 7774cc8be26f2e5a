"""Episode statistics across GPUs.

The batch shards over GPUs as independent lane ranges (no state is shared between envs), so the
only cross-rank traffic of the whole path is this: the 8-element episode-statistics vector each
batch accumulates on its device (sfmi.h: sf_episode_stats), reduced with RCCL over xGMI
(`torch.distributed` backend "nccl", one all-gather) -- or gloo on CPU tensors in the tests.  It stands in for the
trainer's host-side `final_rewards.mean()/median()/min()/max()` and `num_destruction += sum(info)`
(rl/train.py:81,161-164).  One call per log interval; the message is 64 bytes, latency-bound.
"""
import math

import torch

# layout of the vector (include/sfmi.h)
EPISODES, SUM_RETURN, SUM_SQ_RETURN, FORT_KILLS, SHIP_DEATHS, SHOTS, MIN_RETURN, MAX_RETURN = range(8)
INT64_MAX = (1 << 63) - 1
INT64_MIN = -(1 << 63)


def shard_lanes(total_envs, world_size, rank):
    """Contiguous lane range [begin, end) of `rank`: the first `total % world` ranks get one more."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    base, extra = divmod(int(total_envs), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_episode_stats(local, group=None, force=False):
    """Reduce one rank's statistics vector (int64[8], any device) over all ranks: sums for the six counters,
    min / max for the two extremes.  ONE collective -- an all-gather of the 64-byte vectors -- and the fold is
    done locally in rank order (deterministic).  Returns the reduced tensor on the same device."""
    import torch.distributed as dist

    v = torch.as_tensor(local, dtype=torch.int64).clone()
    # force: run the collective for a single rank too (rehearsing the RCCL path on a one-GPU box)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        rows = [torch.empty_like(v) for _ in range(dist.get_world_size(group))]
        dist.all_gather(rows, v.contiguous(), group=group)
        m = torch.stack(rows)
        v = torch.cat([m[:, :6].sum(0), m[:, 6:7].min(0).values, m[:, 7:8].max(0).values])
    return v


def summarize(v):
    """Dict of the quantities the trainer logs (rl/train.py:158-170) from a (reduced) vector."""
    v = [int(x) for x in torch.as_tensor(v).tolist()]
    n = v[EPISODES]
    out = {"episodes": n, "fortress_kills": v[FORT_KILLS], "ship_deaths": v[SHIP_DEATHS], "shots": v[SHOTS]}
    if n > 0:
        mean = v[SUM_RETURN] / n
        var = max(0.0, v[SUM_SQ_RETURN] / n - mean * mean)
        out.update(mean_return=mean, std_return=math.sqrt(var), min_return=v[MIN_RETURN], max_return=v[MAX_RETURN])
    return out
