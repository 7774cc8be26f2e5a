"""The N > 1 path on CPU: two `gloo` processes, each owning a contiguous lane shard (the oracle
plays the shard here -- on GPUs it is an SFVecEnv per rank), accumulate the episode-statistics
vector the kernel keeps on the device, and all-reduce it exactly as bench.py / a trainer does
with RCCL.  The reduced vector must equal the single-process statistics of the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

TOTAL, T = 10, 5295 + 5


def episode_vector(gametype, lanes, actions):
    """What libsfmi accumulates (sfmi.h: sf_episode_stats) for the lanes [lo, hi) of the batch."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O

    lo, hi = lanes
    v = np.array([0, 0, 0, 0, 0, 0, (1 << 63) - 1, -(1 << 63)], np.int64)
    for lane in range(lo, hi):
        env = O.OracleEnv(gametype, spawn_skip=lane)  # spawn_stride = 1
        out = env.replay(actions[:, lane], want_obs=False)
        ends = np.flatnonzero(out["done"])
        start = 0
        for e in ends:
            ret = int(out["reward"][start:e + 1].sum())
            v[0] += 1
            v[1] += ret
            v[2] += ret * ret
            v[3] += int(out["info"][start:e + 1].sum())
            v[4] += int(out["snaps"]["stats"][e][3])
            v[5] += int(out["snaps"]["stats"][e][7])
            v[6] = min(v[6], ret)
            v[7] = max(v[7], ret)
            start = e + 1
    return v


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from spacefortress_amd.stats import reduce_episode_stats, shard_lanes

    rng = np.random.default_rng(5)
    actions = rng.integers(0, 5, (T, TOTAL)).astype(np.uint8)  # same on every rank
    local = episode_vector("youturn", shard_lanes(TOTAL, world, rank), actions)
    red = reduce_episode_stats(torch.from_numpy(local))
    dist.barrier()
    q.put((rank, local.tolist(), red.tolist()))
    dist.destroy_process_group()


def test_shard_lanes():
    from spacefortress_amd.stats import shard_lanes

    for total, world in ((262144, 8), (10, 3), (7, 8), (65536, 1)):
        cuts = [shard_lanes(total, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1
    assert shard_lanes(262144, 8, 3) == (3 * 32768, 4 * 32768)
    with pytest.raises(ValueError):
        shard_lanes(8, 2, 2)


def test_two_rank_gloo_reduction_equals_single_process():
    from spacefortress_amd.stats import summarize

    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    actions = rng.integers(0, 5, (T, TOTAL)).astype(np.uint8)
    whole = episode_vector("youturn", (0, TOTAL), actions)
    assert res[0][2] == res[1][2] == whole.tolist()
    assert res[0][1] != res[1][1]  # the shards really differ
    s = summarize(whole)
    assert s["episodes"] == TOTAL and s["min_return"] <= s["mean_return"] <= s["max_return"]


def test_single_process_reduce_is_identity():
    from spacefortress_amd.stats import reduce_episode_stats, summarize

    v = torch.tensor([3, -30, 400, 1, 20, 50, -20, -5])
    assert torch.equal(reduce_episode_stats(v), v)
    s = summarize(v)
    assert s["mean_return"] == -10 and abs(s["std_return"] ** 2 - (400 / 3 - 100)) < 1e-9
    assert summarize(torch.tensor([0, 0, 0, 0, 0, 0, (1 << 63) - 1, -(1 << 63)]))["episodes"] == 0


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks on its own (rl/train.py:30-32: N workers from
    one command).  SF_BENCH_FORCE_DIST=gloo runs the multi-rank control flow on CPU tensors: nothing is timed, the
    line says "dry_run", but n_gpus, the shard cut, the one-collective reduction, the per-rank evidence gathered from
    every rank and the CPU baseline handed from the launcher to rank 0 are the real code."""
    import json
    import subprocess

    env = dict(os.environ, SF_BENCH_FORCE_DIST="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SF_BENCH_CPU_BASELINE_FILE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--cpu-seconds", "0.4", "--cpu-cores", "2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["steps"] == 20 and j["repeats"] == 100
    assert j["lanes_rank0"] == [0, 65536]
    assert j["max_over_ranks"] == 2.0  # MAX over ranks of (1 + rank)
    es = j["episode_stats"]  # sums of (rank + 1, 10 (rank + 1), ...), min of -5 - rank, max of 7 + rank
    assert es["episodes"] == 3 and es["fortress_kills"] == 1 and es["min_return"] == -6 and es["max_return"] == 8
    # one entry per rank, gathered from the ranks themselves
    assert [x["rank"] for x in j["ranks"]] == [0, 1]
    assert [x["lanes"] for x in j["ranks"]] == [[0, 65536], [65536, 131072]]
    assert [x["block_ms_median"] for x in j["ranks"]] == [1.0, 2.0]
    assert abs(sum(x["value_share"] for x in j["ranks"]) - j["dry_value"]) <= 1e-9 * j["dry_value"] and j["dry_value"] == 2 * 65536 * 20 / 2e-3
    # the CPU baseline of an N > 1 job: timed by the launcher before the ranks start, carried by rank 0's line
    cb = j["cpu_baseline"]
    assert cb is not None and cb["cores"] == 2 and cb["value"] > 1e4 and cb["kind"] in ("reference", "port")
    # a WORLD_SIZE that disagrees with --gpus is an error, not a silent single shard
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE" in r2.stderr


def test_a_rank_that_dies_takes_the_job_down_at_once():
    """the launcher used to wait for rank 0 first, so a rank that died at start-up left its peers in
    init_process_group / a collective until the store's timeout (tens of minutes).  All ranks are polled now: the first
    failure stops the others and its exit code is the job's."""
    import subprocess
    import time

    env = dict(os.environ, SF_BENCH_FORCE_DIST="gloo", SF_BENCH_TEST_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SF_BENCH_CPU_BASELINE_FILE"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 60  # not a rendezvous timeout
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # and no line from a half-run job


def _bench_env(**extra):
    env = dict(os.environ, SF_BENCH_FORCE_DIST="gloo", **extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "SF_BENCH_CPU_BASELINE_FILE"):
        env.pop(k, None)
    return env


def test_world_of_eight_dry_run():
    """The driver's 8-GPU command rehearsed on CPU tensors (gloo): bench.py starts eight fresh ranks, polls them, hands the
    CPU baseline from the launcher to rank 0 through a file; the line carries eight `ranks` entries gathered from the ranks
    themselves -- eight contiguous shards of one 524 288-lane batch, eight distinct device ids -- and the size of the
    process group the collective ran in."""
    import json
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
                        "--cpu-seconds", "0.3", "--cpu-cores", "2"],
                       env=_bench_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 8 and j["rccl_world"] == 8
    assert [x["rank"] for x in j["ranks"]] == list(range(8))
    assert [x["lanes"] for x in j["ranks"]] == [[65536 * k, 65536 * (k + 1)] for k in range(8)]
    assert len({(x["pci_bus_id"], x["uuid"]) for x in j["ranks"]}) == 8
    assert j["max_over_ranks"] == 8.0
    # what a first real 8-GPU line will be judged on: `value` is the SUM of the ranks' shares -- every rank's own
    # count of env-steps over the job's clock (MAX over ranks: 8 ms here) --, and the ranks' lanes tile one batch of N n lanes
    assert [x["env_steps"] for x in j["ranks"]] == [65536 * 20] * 8
    assert abs(sum(x["value_share"] for x in j["ranks"]) - j["dry_value"]) <= 1e-9 * j["dry_value"]
    assert abs(j["dry_value"] - 8 * 65536 * 20 / 8e-3) <= 1e-6 * j["dry_value"]
    edges = sorted(tuple(x["lanes"]) for x in j["ranks"])
    assert edges[0][0] == 0 and edges[-1][1] == 8 * 65536 and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    es = j["episode_stats"]  # sums over ranks of (rank + 1, ...); min of -5 - rank, max of 7 + rank
    assert es["episodes"] == 36 and es["min_return"] == -12 and es["max_return"] == 14
    cb = j["cpu_baseline"]
    assert cb is not None and cb["cores"] == 2 and cb["single_core"]["value"] > 1e4
    assert cb["config0"]["steps"] == 1000 and cb["config0"]["value"] > 1e3
    assert cb["subproc_vecenv"]["procs"] == 2 and cb["subproc_vecenv"]["value"] > 100


def test_dead_rank_five_of_eight_takes_the_job_down():
    import subprocess
    import time

    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline"], env=_bench_env(SF_BENCH_TEST_FAIL_RANK="5"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 90
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_two_ranks_on_one_device_fail_the_job():
    """An N-GPU line must be evidence of N GPUs: ranks that report the same device id end the job with a non-zero code
    instead of a line that looks like bad scaling."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline"], env=_bench_env(SF_BENCH_TEST_SAME_DEVICE="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 4, (r.returncode, r.stderr[-1000:])
    assert "distinct devices" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
