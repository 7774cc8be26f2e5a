"""The sharded batch (BASELINE.json configs[3]: 262 144 youturn envs over 8 GPUs, per-lane auto-reset) and the spawn
stream of its highest lanes.

A job of W ranks is ONE batch of W * n lanes cut into contiguous shards (spacefortress_amd.stats.shard_lanes): rank r
creates its shard with spawn_stride = 1 and spawn_skip = r * n, so lane i of rank r is lane r * n + i of the whole batch
and continues the libc rand() stream from there (SRC/game.cpp:133-149), which is what OracleEnv(spawn_skip=lane) does.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from sfcompare import compare_state
from test_gpu_parity import obs_close, run_device

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    from spacefortress_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsfmi.so not built: the GPU tests never fall back"
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


def test_top_lanes_of_the_metric_batch_continue_the_libc_stream(sfa, oracle_mod):
    """65 536 lanes, spawn_stride 1 (bench.py's batch): the LAST lanes start at entries 65 500 .. 65 535 of the accepted-
    spawn sequence and must walk on from there through their respawns -- a 65 536-entry table would wrap them to entry
    0 at their first death.  2 400 steps = about 30 respawns per lane, every output of every step against the oracle."""
    O = oracle_mod
    n, T, lo = 65536, 2400, 65500
    rng = np.random.default_rng(11)
    ring = rng.integers(0, 5, (64, n)).astype(np.uint8)
    env = sfa.SFVecEnv(n, gametype="youturn", spawn_stride=1, reuse_buffers=False)
    orc = O.OracleVecEnv("youturn", n - lo, spawn_skip=lo, spawn_stride=1)
    dev_ring = torch.from_numpy(ring).to(env.device)
    k = n - lo
    obs = torch.empty((T, k, env.obs_dim), dtype=env.obs_dtype, device=env.device)
    rew = torch.empty((T, k), dtype=torch.int32, device=env.device)
    env.reset()
    orc.reset()
    for t in range(T):
        o, r, d, i = env.step_tensors(dev_ring[t % 64])
        obs[t].copy_(o[lo:])
        rew[t].copy_(r[lo:])
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    for t in range(T):
        oo, orw, od, oi = orc.step(ring[t % 64, lo:].astype(np.int32))
        assert np.array_equal(rew[t], orw), (t, np.flatnonzero(rew[t] != orw)[:5])
        assert obs_close(obs[t], oo, False).all(), t
    sd = env.state_dict()
    snaps = orc.snapshots()
    assert (snaps["stats"][:, 3] >= 10).all()  # every one of them has respawned many times
    bad = compare_state(sd, snaps, lanes=np.arange(lo, n))
    assert not bad, bad
    # the cursors really are beyond the old table's end
    assert (sd["spawn_cursor"][-20:] > 65536).all()
    env.close()


def test_a_spawn_table_that_ends_before_the_last_lane_is_refused(sfa):
    with pytest.raises(ValueError):
        sfa.SFVecEnv(4096, gametype="youturn", spawn_stride=64, spawn_table_len=65536)  # last start 262 080
    env = sfa.SFVecEnv(4096, gametype="youturn", spawn_stride=64)  # default: sized past the last lane
    env.close()


def test_two_shards_equal_one_batch(sfa):
    """Ranks' shards with spawn_skip = first lane are bit-identical to the corresponding halves of the single batch."""
    n, T = 2048, 700
    rng = np.random.default_rng(5)
    acts = rng.integers(0, 5, (T, 2 * n)).astype(np.uint8)
    whole = sfa.SFVecEnv(2 * n, gametype="youturn", spawn_stride=1)
    ow = run_device(whole, acts)
    sw = whole.state_dict()
    whole.close()
    from spacefortress_amd.stats import shard_lanes

    for rank in range(2):
        a, b = shard_lanes(2 * n, 2, rank)
        sh = sfa.SFVecEnv(b - a, gametype="youturn", spawn_stride=1, spawn_skip=a)
        os_ = run_device(sh, acts[:, a:b])
        for x, y in zip(ow, os_):
            assert np.array_equal(x[:, a:b], y)
        ss = sh.state_dict()
        for k in ss:
            assert np.array_equal(ss[k], sw[k][..., a:b]), k
        sh.close()


def test_config4_shard_through_the_auto_reset(sfa, oracle_mod):
    """BASELINE.json configs[3], one rank's share: rank 3 of 8, youturn, 32 768 lanes (lanes 98 304 .. 131 071 of the
    262 144-lane batch), 5 300 steps so that every lane finishes its episode at step 5 295 and is auto-reset in the
    kernel.  Sampled lanes against the oracle for the whole run (obs / reward / done / info and the final state, which
    is five steps into the SECOND episode); invariants and the device-side episode accumulators over all lanes."""
    O = oracle_mod
    from spacefortress_amd.stats import shard_lanes

    lane0, lane1 = shard_lanes(262144, 8, 3)
    n, T = lane1 - lane0, 5300
    assert (lane0, n) == (98304, 32768)
    rng = np.random.default_rng(4)
    ring = rng.integers(0, 5, (64, n)).astype(np.uint8)
    pick = np.sort(rng.choice(n, 96, replace=False))
    pick[0], pick[-1] = 0, n - 1
    env = sfa.SFVecEnv(n, gametype="youturn", spawn_stride=1, spawn_skip=lane0)
    dev_ring = torch.from_numpy(ring).to(env.device)
    dpick = torch.from_numpy(pick).to(env.device)
    obs = torch.empty((T, len(pick), env.obs_dim), dtype=env.obs_dtype, device=env.device)
    rew = torch.empty((T, len(pick)), dtype=torch.int32, device=env.device)
    dsum = torch.zeros(T, dtype=torch.int64, device=env.device)
    isum = torch.zeros((), dtype=torch.int64, device=env.device)
    ret = torch.zeros(n, dtype=torch.int64, device=env.device)
    fin = torch.zeros(n, dtype=torch.int64, device=env.device)
    env.reset()
    for t in range(T):
        o, r, d, i = env.step_tensors(dev_ring[t % 64])
        obs[t] = o[dpick]
        rew[t] = r[dpick]
        dsum[t] = d.sum()
        isum += i.sum()
        ret += r
        fin = torch.where(d.bool(), ret, fin)
        ret = torch.where(d.bool(), torch.zeros_like(ret), ret)
    dsum = dsum.cpu().numpy()
    assert dsum[5294] == n and dsum.sum() == n            # every lane is done at step 5 295, and only then
    st = env.episode_stats()
    fin = fin.cpu().numpy()
    assert st[0] == n and st[1] == fin.sum() and st[2] == (fin ** 2).sum() and st[6] == fin.min() and st[7] == fin.max()
    sd = env.state_dict()
    assert (sd["time"] == 34 * (T - 5295)).all() and (sd["points"] >= 0).all()
    assert np.array_equal(sd["stats"][3], sd["stats"][0] + sd["stats"][1] + sd["stats"][2])
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    snaps = []
    for j, lane in enumerate(pick):
        o = O.OracleVecEnv("youturn", 1, spawn_skip=int(lane0 + lane))
        o.reset()
        for t in range(T):
            oo, orw, od, oi = o.step(ring[t % 64, lane:lane + 1].astype(np.int32))
            assert rew[t, j] == orw[0], (lane, t)
            assert bool(od[0]) == (t == 5294)
            assert obs_close(obs[t, j], oo[0], False).all(), (lane, t)
        snaps.append(o.snapshots()[0])
    bad = compare_state(sd, np.array(snaps), lanes=pick)
    assert not bad, bad
    env.close()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _bench_rccl(extra):
    env = dict(os.environ, SF_BENCH_FORCE_DIST="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SF_BENCH_CPU_BASELINE_FILE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--envs", "4096", "--no-cpu-baseline",
                        "--rollout-k", "0", "--image-envs", "0", "--numpy-api", "0", "--no-configs", "--steady-seconds", "0.3",
                        "--kernel-timing-launches", "10"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_rccl_path_on_one_rank():
    """bench.py's multi-rank control flow over RCCL (backend "nccl"), rehearsed with the ONE rank this box has:
    process group on the device, NCCL barriers around every timed block, the MAX all-reduce of the block time and the
    all-gather of the 64-byte statistics vector.  (Two ranks cannot share a device under RCCL; the two-rank flow runs
    on gloo in tests/test_stats_gloo.py.)"""
    j = _bench_rccl(["--steps", "5400", "--warmup", "10"])
    assert j["n_gpus"] == 1 and j["steps"] == 5400 and j["repeats"] == 1 and j["launch"].startswith("loop")
    es = j["episode_stats"]  # every lane finished whole episodes inside the run, in step with one another
    assert es["episodes"] >= 2 * 4096 and es["episodes"] % 4096 == 0
    assert j["value"] > 1e8 and j["stats_reduce_us"] > 0


def test_the_drivers_scale_command_on_one_rank():
    """What the driver runs for its scaling table -- `bench.py --gpus N --steps 20 --warmup 5` under one rank per GPU
    with an initialised RCCL process group -- takes the HIP-GRAPH path (K <= 512), which VERDICT r2 found never
    executed next to a live communicator: the graphs are captured before init_process_group (thread-local capture
    mode), replayed between NCCL barriers.  The line must say who ran (`ranks`: device, PCI bus id, per-rank block time)
    and carry the action-generation figure."""
    j = _bench_rccl(["--steps", "20", "--warmup", "5"])
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["repeats"] == 100 and j["launch"].startswith("hip_graph")
    assert j["value"] > 1e8 and 0 < j["ms_per_step"] < 0.1
    assert j["value_with_action_gen"] > 1e8 and j["action_gen"]["ms_per_step"] > 0
    assert j["loop_issue"]["value"] > 1e7
    (rk,) = j["ranks"]
    assert rk["rank"] == 0 and rk["device_index"] == 0 and rk["lanes"] == [0, 4096] and rk["block_ms_median"] > 0
    assert rk["pci_bus_id"] is None or len(rk["pci_bus_id"]) >= 7
    assert j["episode_stats"]["episodes"] >= 4096  # the steady-state loop plays whole episodes: the all-gather carries counts
