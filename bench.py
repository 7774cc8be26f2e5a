#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Space Fortress env.step() on MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 100
    python bench.py --gpus N ...          N > 1 without WORLD_SIZE: bench.py starts the N ranks itself (fresh child
                                          processes, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; rl/train.py:30-32 starts
                                          its N workers from one command too)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one sf_step launch: one 34 ms game tick for every env of the batch, fused
with action decoding, reward shaping, the 19-float observation and auto-reset.  Workload:
BASELINE.json's metric config -- youturn, 65 536 envs per GPU, features obs (f32), uniform
random discrete actions pre-generated in HBM (a ring of 64 action batches).  N > 1: one
process per GPU, each with its own 65 536-lane shard (weak scaling, no data-path
collective); RCCL all-gathers the 8-element episode-statistics vector once, after the
timed region.

The JSON line also carries
  roofline      the step kernel against the HBM roofline: algorithmic bytes per launch
                (464 B/env-step youturn, 408 autoturn) / mean launch duration,
                measured here with HIP events on the launch stream;
  cpu_baseline  the REAL reference engine (oracle/_ref, bare C++ tick loop) -- or the C
                restatement if that build is absent -- timed on the host cores, one process
                per core as the reference itself parallelises, ~10 s sample.  Runs before
                the GPU is touched (N > 1: in the launcher before the ranks start, or on
                rank 0 before it joins the process group).  Reported baseline, not the target;
  value_with_action_gen   the same timed blocks with the actions DRAWN INSIDE the launch
                (sf_step_sampled: Philox4x32-10 per lane and tick, SURVEY 8d "action
                generation on device ... report both"; rl/train.py:76-80 has the policy's
                sample there);
  configs       every other BASELINE.json configuration, each a few HIP-graph blocks on
                this GPU (N = 1 only): youturn 4 096, autoturn 65 536, youturn 32 768
                (configs[3]'s per-GPU share), youturn 262 144, and the image config;
  ranks         per rank: device index, PCI bus id, its own median block time -- one
                all-gather, so that the line shows N distinct GPUs.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES = {"youturn": 464, "autoturn": 408, "test-youturn": 464, "test-autoturn": 408}
IMAGE_ALGO_BYTES = 7448  # one new 84x84 frame instead of the feature row
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
METRIC = "env-steps/sec (whole node), youturn random-action rollout @65536 envs/GPU"


def cpu_baseline(gametype, seconds, cores):
    """One OS process per core, each stepping one env with random actions (oracle/cpu_bench.py)."""
    kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsfref.so")) else "port"
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--kind", kind, "--gametype", gametype,
           "--seconds", str(seconds)]
    procs = [subprocess.Popen(cmd + ["--seed", str(100 + i)], stdout=subprocess.PIPE, text=True) for i in range(cores)]
    total, kinds = 0.0, set()
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            return None
        r = json.loads(out.strip().splitlines()[-1])
        total += r["steps"] / r["seconds"]
        kinds.add(r["kind"])
    kind = kinds.pop() if len(kinds) == 1 else "port"
    what = ("reference C++ engine (oracle/_ref: Game::pressKey/releaseKey + stepOneTick(34) loop, new Game at game over)"
            if kind == "reference" else "C restatement of the engine (oracle/sf_oracle.c), same loop")
    out = {"value": total, "unit": "env-steps/s", "cores": cores, "kind": kind,
           "sample": "%s, %s, uniform random actions, one process per core x %d, %.1f s each; "
                     "bare engine only (no Python wrapper, no IPC)" % (what, gametype, cores, seconds)}
    # the bare engine on ONE core as well
    try:
        r = subprocess.run(cmd + ["--seed", "99", "--seconds", str(min(3.0, seconds))], stdout=subprocess.PIPE, text=True, timeout=120)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        out["single_core"] = {"value": j["steps"] / j["seconds"], "unit": "env-steps/s", "cores": 1, "kind": j["kind"],
                              "sample": "the same loop in ONE process, %.1f s" % j["seconds"]}
    except Exception as e:  # the baseline is informative, never fatal
        out["single_core"] = {"error": str(e)}
    # the reference's actual shape: SubprocVecEnv (one process per env, Pipe IPC, Python wrapper, rl/train.py:30-32) with its
    # default 16 processes (rl/arguments.py:21-22) -- capped at this box's share -- and with one process per available core
    def subproc(procs, secs):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "subproc_bench.py"), "--procs", str(procs),
                                "--gametype", gametype, "--seconds", str(secs)], stdout=subprocess.PIPE, text=True, timeout=120)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            return {"value": j["steps"] / j["seconds"], "unit": "env-steps/s", "procs": j["procs"],
                    "sample": "SubprocVecEnv-shaped harness (oracle/subproc_bench.py): one process per env, "
                              "multiprocessing.Pipe, per-env wrapper step + features obs, reset on done; "
                              "the C restatement inside each worker"}
        except Exception as e:
            return {"error": str(e)}
    out["subproc_vecenv"] = subproc(min(16, cores), max(0.5, seconds / 2))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    avail = min(avail, 64)  # (a GPU box allows a job a bounded number of processes)
    if avail != min(16, cores):
        out["subproc_vecenv_all_cores"] = dict(subproc(avail, max(0.5, seconds / 4)), host_cores=avail,
                                               note="num_processes = the cores this job may use (os.sched_getaffinity, at most 64)")
    # BASELINE.json configs[0]: youturn, 1 env on the CPU through the engine's per-call entry points, 1000-step random rollout
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--kind", kind, "--gametype", gametype,
                            "--rollout-steps", "1000"], stdout=subprocess.PIPE, text=True, timeout=120)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        out["config0"] = {"name": "configs[0]: %s, 1 env on the CPU, 1000-step random rollout" % gametype,
                          "value": j["steps"] / j["seconds"], "unit": "env-steps/s", "steps": j["steps"], "ms_total": j["seconds"] * 1e3,
                          "kind": j["kind"],
                          "sample": "ONE env stepped from Python one call at a time -- press_key / release_key x 4, step_one_tick(34), "
                                    "is_game_over, as SSF_Env.step does (ENV:208-253) -- through ctypes on the %s; actions "
                                    "numpy RandomState(0).randint; best of 5 runs" %
                                    ("reference C++ engine (oracle/_ref)" if j["kind"] == "reference" else "C restatement")}
    except Exception as e:
        out["config0"] = {"error": str(e)}
    return out


def baseline_cores(args):
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    # a one-GPU box owns 16 host cores of the node whatever os.cpu_count() says
    return args.cpu_cores or max(1, min(share, 16))


def committed_profile(gametype, envs, obs_type):
    """What the committed rocprofv3 runs of this same command measured (profiles/step_kernel_latest.json,
    written by tools/pmc_report.py + tools/trace_report.py): HBM bytes per launch from the two --pmc passes
    and the kernel-trace mean duration.  bench.py cannot observe either itself: these two numbers are
    REPLAYED from the profile of the named kernel version, and labelled so.  {} when no profile matches."""
    try:
        rep = json.load(open(os.path.join(ROOT, "profiles", "step_kernel_latest.json")))
    except Exception:
        return {}
    w = rep.get("workload", {})
    if (w.get("gametype"), w.get("envs_per_gpu"), w.get("obs_type")) != (gametype, envs, obs_type):
        return {}
    # the profile belongs to ONE build of the library (sf_build_id = the hash of its sources, stored by
    # tools/profile_version.sh): replayed under any other build it is marked stale
    try:
        from spacefortress_amd import _lib
        rep["loaded_build_id"] = _lib.lib().sf_build_id().decode()
    except Exception:
        rep["loaded_build_id"] = None
    rep["stale"] = not (rep.get("sf_build_id") and rep.get("sf_build_id") == rep["loaded_build_id"])
    return rep


def committed_image_profile(envs):
    """The render kernel's rocprofv3 --kernel-trace mean at the image config (profiles/image_kernel_latest.json, written by
    tools/profile_version.sh from the same run as profiles/rNN_render_kernel_stats_*.csv), with the same stale rule."""
    try:
        rep = json.load(open(os.path.join(ROOT, "profiles", "image_kernel_latest.json")))
    except Exception:
        return {}
    if rep.get("envs") != envs:
        return {}
    try:
        from spacefortress_amd import _lib
        rep["loaded_build_id"] = _lib.lib().sf_build_id().decode()
    except Exception:
        rep["loaded_build_id"] = None
    rep["stale"] = not (rep.get("sf_build_id") and rep.get("sf_build_id") == rep["loaded_build_id"])
    return rep


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, args):
    """`python bench.py --gpus N` on its own: start the N ranks as FRESH child processes (nothing here has touched
    the GPU, and nothing is re-exec'ed), one per GPU, with the torch.distributed environment set; rank 0 prints the
    JSON line.  The reference starts its N workers from one command as well (rl/train.py:30-32).
    The CPU baseline is timed HERE, before the ranks exist (this process never touches a GPU), and handed to rank 0
    through a file.  All children are polled together: the first one that fails takes the others down with it (a rank
    that dies at start-up would otherwise leave its peers in a collective until the NCCL timeout), and the whole job
    has a deadline."""
    env_common = dict(os.environ)
    tmp = None
    dry = os.environ.get("SF_BENCH_FORCE_DIST", "") == "gloo"
    if not args.no_cpu_baseline:
        base = cpu_baseline(args.gametype, args.cpu_seconds, baseline_cores(args))
        fd, tmp = tempfile.mkstemp(prefix="sf_bench_cpu_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(base, f)
        env_common["SF_BENCH_CPU_BASELINE_FILE"] = tmp
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(env_common, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    deadline = time.time() + float(os.environ.get("SF_BENCH_TIMEOUT", "300" if dry else "1500"))
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed:
                rc = failed[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                sys.stderr.write("bench.py: the ranks did not finish in time; stopping them\n")
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # a failed / late job: nobody is left waiting in a collective
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 5
        for p in procs:
            while p.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
        if tmp:
            try:
                os.unlink(tmp)
            except OSError:
                pass
    return rc


def device_identity(torch, local_rank):
    p = torch.cuda.get_device_properties(local_rank)
    dom, bus, devn = (getattr(p, k, -1) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    return {"device_index": local_rank, "name": p.name,
            "pci_bus_id": "%04x:%02x:%02x.0" % (dom, bus, devn) if bus >= 0 else None,
            "uuid": str(getattr(p, "uuid", "")) or None}


def whole_job_value(ranks, elapsed_s):
    """`value` of an N-rank line: the units all ranks processed in a timed block (each rank's own report, `env_steps`) over the
    block's time (barrier-to-barrier, MAX over ranks).  Leaves every rank's share at that clock in its entry (`value_share`):
    the line's value is their sum, and the ranks' lane ranges must tile [0, N n) -- one batch cut into contiguous shards."""
    for r in ranks:
        r["value_share"] = r["env_steps"] / elapsed_s
    lanes = sorted(tuple(r["lanes"]) for r in ranks)
    if lanes[0][0] != 0 or any(a[1] != b[0] for a, b in zip(lanes, lanes[1:])):
        raise SystemExit("bench.py: the ranks' lane ranges do not tile one batch: %r" % (lanes,))
    return sum(r["value_share"] for r in ranks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12000, help="timed launches per block (default spans two episode rollovers: 5295 steps each)")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed blocks of --steps launches each, every one bracketed by barrier + synchronize; the line "
                         "reports the MEDIAN block (0 = enough blocks for about 2000 launches in all, at most 101)")
    ap.add_argument("--launch", choices=("auto", "loop", "graph"), default="auto",
                    help="how a timed block's K sf_step launches are issued: one by one from Python (loop) or as ONE HIP graph "
                         "captured once and replayed per block (graph); auto = graph for K <= 512.  A block of a few launches "
                         "issued one by one starts on an idle GPU with the host barely ahead of it (5 us per call against a "
                         "6.5 us kernel): the first launches wait for their packets.  The line says which (`launch`) and, for "
                         "a graph, carries the loop-issued figure beside it (`loop_issue`)")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--gametype", default="youturn")
    ap.add_argument("--obs-type", default="features")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-cores", type=int, default=0, help="baseline processes (0 = this box's share, at most 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-timing-launches", type=int, default=200)
    ap.add_argument("--steady-seconds", type=float, default=3.0,
                    help="length of the steady-state loop outside the timed blocks (back-to-back launches from the Python "
                         "loop): long enough for an external GPU-utilisation sampler to see it, and for every lane to "
                         "finish episodes, so that the statistics all-gather carries something")
    ap.add_argument("--rollout-k", type=int, default=64, metavar="K",
                    help="also time the fused open-loop path (sf_rollout: K ticks per launch, all actions known up "
                         "front, same per-tick outputs); reported as rollout_fused, never as value; 0 = skip")
    ap.add_argument("--numpy-api", type=int, default=200, metavar="K",
                    help="also time K steps through the host-buffer (numpy) API -- what rl/train.py:79-80 calls: actions "
                         "H2D, results D2H every step -- the PCIe-inclusive rate, reported as host_api, never as value; 0 = skip")
    ap.add_argument("--image-envs", type=int, default=16384, metavar="N",
                    help="also time N envs stepping with the image observation (BASELINE cfg 5: sf_step + sf_render, "
                         "uint8 [N,1,84,84] per step); reported as image_obs, never as value; 0 = skip")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (`configs`)")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling runs (tools/profile_version.sh): nothing but the warm-up and the timed blocks of sf_step launches, "
                         "so that a kernel trace of the run averages those launches alone (no sampled-action blocks, no "
                         "loop-issued blocks, no steady-state loop, no extras)")
    args = ap.parse_args()
    if args.timed_only:
        args.rollout_k = args.numpy_api = args.image_envs = 0
        args.no_configs = args.no_cpu_baseline = True
        args.steady_seconds = 0.0
        args.kernel_timing_launches = 1

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not os.environ.get("SF_BENCH_NO_SPAWN"):
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or drop WORLD_SIZE and let "
                 "bench.py start the ranks)" % (args.gpus, world))
    # SF_BENCH_FORCE_DIST=gloo: the multi-rank control flow on CPU tensors (no GPU: nothing is timed, the line
    # carries "dry_run": true); =1 / nccl: the RCCL path, also for a single rank
    force = os.environ.get("SF_BENCH_FORCE_DIST", "")
    dry = force == "gloo"

    if os.environ.get("SF_BENCH_TEST_FAIL_RANK") == str(rank):  # tests/test_stats_gloo.py: a rank that dies at start-up
        sys.exit(3)
    base = None
    if rank == 0 and not args.no_cpu_baseline:
        handed = os.environ.get("SF_BENCH_CPU_BASELINE_FILE")
        if handed and os.path.exists(handed):  # timed by the launcher before the ranks started
            base = json.load(open(handed))
        else:
            # before anything initialises the GPU in this process (children are plain CPU processes); under a foreign
            # launcher (torchrun) the other ranks wait for rank 0 in init_process_group meanwhile
            base = cpu_baseline(args.gametype, args.cpu_seconds, baseline_cores(args))

    import numpy as np
    import torch

    from spacefortress_amd.stats import reduce_episode_stats, shard_lanes, summarize

    dist = None
    if world > 1 or force:
        import torch.distributed as dist

    n = args.envs
    # the job is ONE batch of world * n lanes cut into contiguous shards: rank r's lane i is lane r*n + i of the whole
    # batch and takes that stretch of the spawn stream (spawn_stride 1), so N ranks equal one N*n-lane batch
    lane0, lane1 = shard_lanes(n * world, world, rank)
    repeats = args.repeats or max(1, min(101, -(-2000 // max(1, args.steps))))

    def gather_ranks(mine):
        if dist is None:
            return [mine]
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        return rows

    if dry:
        dist.init_process_group("gloo")
        local = torch.tensor([rank + 1, 10 * (rank + 1), 100, rank, 2, 3, -5 - rank, 7 + rank], dtype=torch.int64)
        stats = reduce_episode_stats(local)
        tt = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        # (dry run: a stand-in device id per rank; SF_BENCH_TEST_SAME_DEVICE makes two ranks claim the same one)
        fake = 0 if os.environ.get("SF_BENCH_TEST_SAME_DEVICE") else rank
        ranks = gather_ranks({"rank": rank, "device_index": None, "pci_bus_id": "dry:%02x" % fake, "uuid": "dry-%d" % fake,
                              "block_ms_median": 1.0 + rank, "lanes": [lane0, lane1], "env_steps": (lane1 - lane0) * args.steps})
        whole_job_value(ranks, float(tt.item()) * 1e-3)  # (the real path's arithmetic on the stand-in clock: MAX over ranks)
        if len({(r["pci_bus_id"], r["uuid"]) for r in ranks}) != world or dist.get_world_size() != world:
            sys.stderr.write("bench.py: %d ranks but not %d distinct devices\n" % (world, world))
            sys.exit(4)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "dry_run": True, "value": None, "dry_value": whole_job_value(ranks, float(tt.item()) * 1e-3),
                              "n_gpus": world, "steps": args.steps,
                              "rccl_world": dist.get_world_size(),
                              "warmup": args.warmup, "repeats": repeats, "lanes_rank0": [lane0, lane1],
                              "max_over_ranks": float(tt.item()), "ranks": ranks, "cpu_baseline": base,
                              "episode_stats": summarize(stats)}))
        dist.destroy_process_group()
        return

    from spacefortress_amd import SFVecEnv

    # (device_count() does not initialise the GPU on this image: a job with more ranks than GPUs ends here, at once,
    #  and the launcher stops the other ranks)
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d has no GPU (%d visible for --gpus %d)" % (local_rank, torch.cuda.device_count(), world))
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def sync():
        torch.cuda.synchronize()

    def capture(fn, count):
        """`count` calls of fn(k) as ONE HIP graph (sf_step is a pure stream operation: tests/test_gpu_capi_native.py::
        test_steps_can_be_captured_in_a_hip_graph).  Thread-local capture mode: other threads of the process (a
        process group's watchdog) may make HIP calls meanwhile."""
        sync()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                for k in range(count):
                    fn(k)
        torch.cuda.current_stream(dev).wait_stream(side)
        sync()
        graph.replay()  # one untimed replay: the first one uploads the graph
        sync()
        return graph

    env = SFVecEnv(n, gametype=args.gametype, obs_type=args.obs_type, device=dev, spawn_stride=1,
                   spawn_skip=lane0, reuse_buffers=True)
    env.seed_actions(1234, first_lane=lane0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    ring = 64
    actions = torch.randint(0, env.n_actions, (ring, n), device=dev, dtype=torch.uint8, generator=g)
    env.reset()

    act_rows = [actions[k] for k in range(ring)]  # the views made once: the launch loop must not be what is measured
    step = env.step_tensors
    step_sampled = env.step_sampled
    for t in range(args.warmup):
        step(act_rows[t % ring])
    K = args.steps
    use_graph = args.launch == "graph" or (args.launch == "auto" and K <= 512)
    # Every graph is captured BEFORE the process group exists: a capture next to a live NCCL communicator and its watchdog
    # thread is the classic "operation not permitted when stream is capturing"; replays are ordinary launches.
    # Every block replays the same K action rows on the state the previous block left, so episodes still progress.
    graph = capture(lambda k: step(act_rows[k % ring]), K) if use_graph else None
    graph_sampled = capture(lambda k: step_sampled(), K) if use_graph else None

    if dist is not None:
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed_blocks(graph_, launch, reps):
        """reps timed blocks of EXACTLY K launches, barrier + synchronize on both sides, nothing else inside; returns
        (block seconds -- max over ranks --, this rank's own block seconds, launch periods from HIP events, which blocks
        carried the events).  Every OTHER block (the odd ones; the only one, if there is one) carries a pair of HIP events
        around its launches: they measure the launch period on the GPU's clock for `roofline`, and they cost the block 7 us
        of its own (two barrier packets on the stream and their completion: tools/block_probe.py, 155.5 against 148.0 us at
        K = 20) -- so `value` comes from the blocks WITHOUT them, which hold the K launches between the brackets and nothing
        else, and the event blocks' own median is reported beside it (block_ms.with_events_median)."""
        blocks, own, periods, with_ev = [], [], [], []
        for rep in range(reps):
            evs = reps == 1 or rep % 2 == 1
            barrier()
            t0 = time.perf_counter()
            if graph_ is not None:
                if evs:
                    ev0.record()  # HIP events around the replay: K launch periods + the graph's own launch latency
                graph_.replay()
                if evs:
                    ev1.record()
            else:
                launch(0)
                # HIP events on the launch stream: the first one BEHIND the first launch (it completes when that kernel
                # does), the second behind the last, so that K - 1 launch periods are measured on the GPU's clock and the
                # host's latency in front of an idle GPU is not counted as kernel time (the wall clock counts everything)
                if evs:
                    ev0.record()
                for t in range(1, K):
                    launch(t)
                if evs:
                    ev1.record()
            sync()  # the contract's synchronize: this rank's K launches are done, its clock stops
            elapsed = time.perf_counter() - t0
            own.append(elapsed)
            if dist is not None:
                # the closing barrier of the bracket, and the MAX over ranks: the starts were aligned by the opening barrier,
                # so the slowest rank's (stop - start) is the job's time for the block; the collective's own latency (tens
                # of microseconds of RCCL against a 160 us block at K = 20) is not part of anybody's K launches
                barrier()
                tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                elapsed = float(tt.item())
            blocks.append(elapsed)
            with_ev.append(evs)
            if evs:
                periods.append(ev0.elapsed_time(ev1) / (K if graph_ is not None else max(1, K - 1)))
        return blocks, own, periods, with_ev

    def pick(blocks, with_ev):
        """(index of the median block among those without events -- all of them if every block carried events --, sorted indices)"""
        idx = [r for r in range(len(blocks)) if not with_ev[r]] or list(range(len(blocks)))
        idx.sort(key=lambda r: blocks[r])
        return idx[len(idx) // 2], idx

    tpos = [args.warmup]

    def launch_ring(t):
        step(act_rows[(tpos[0] + t) % ring])

    blocks, own_blocks, periods, with_ev = timed_blocks(graph, launch_ring, repeats)
    tpos[0] += K * repeats
    env.check_state()  # after the timed blocks (every rank its own shard)
    med, order = pick(blocks, with_ev)
    elapsed = blocks[med]
    ev_blocks = sorted(blocks[r] for r in range(repeats) if with_ev[r])
    # mean launch-to-launch time of sf_step_kernel over the blocks that carried the HIP events (their median; end of the first
    # launch to end of the last): the launches are back to back on one stream, so this is the kernel duration plus the
    # dependent-launch gap -- a launch PERIOD
    region_ms = sorted(periods)[len(periods) // 2]

    # ---- the same blocks with the actions drawn inside the launch (sf_step_sampled)
    s_reps = 1 if args.timed_only else repeats
    s_blocks, _, s_periods, s_ev = timed_blocks(graph_sampled, lambda t: step_sampled(), s_reps)
    s_med, _ = pick(s_blocks, s_ev)
    s_period = sorted(s_periods)[len(s_periods) // 2]
    # ---- ... and, when the blocks above were HIP graphs, a few blocks issued one by one from Python: the per-call host path
    loop_issue = None
    if graph is not None and not args.timed_only:
        l_blocks, _, l_periods, l_ev = timed_blocks(None, launch_ring, max(4, repeats // 8))
        l_med, _ = pick(l_blocks, l_ev)
        loop_issue = {"value": float(n) * K * world / l_blocks[l_med], "ms_per_step": l_blocks[l_med] / K * 1e3,
                      "launch_period_ms": sorted(l_periods)[len(l_periods) // 2], "blocks": len(l_blocks),
                      "note": "the same K-launch blocks issued one by one from Python (one env.step_tensors call per launch) "
                              "instead of one graph replay: what a caller's per-step loop pays at this block length"}

    # ---- the launch period in steady state, outside the timed blocks: a short block starts on an idle GPU (first-launch
    #      / graph-launch latency, packets barely ahead of the kernels), so its average reads above what the kernel
    #      sustains.  Back-to-back launches from the Python loop for --steady-seconds, in chunks between HIP events: every
    #      lane plays through whole episodes meanwhile (the statistics below are not zeros), and a utilisation sampler
    #      outside this process gets to see the GPU busy.
    chunk = 64 if args.timed_only else 4096
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step(act_rows[0])
    sync()
    n_steady, steady_total_ms, t_begin = 0, 0.0, time.perf_counter()
    while n_steady < (chunk if args.timed_only else max(K, 2000)) or time.perf_counter() - t_begin < args.steady_seconds:
        s0.record()
        for t in range(chunk):
            step(act_rows[t % ring])
        s1.record()
        sync()
        steady_total_ms += s0.elapsed_time(s1)
        n_steady += chunk
    steady_ms = steady_total_ms / n_steady
    env.check_state()  # (raises on the sticky device error word: nothing below is reported for a batch that overflowed)

    # ---- the only collective of the path, outside the timed blocks and timed on its own: 64 bytes over RCCL
    sync()
    ts = time.perf_counter()
    stats = torch.from_numpy(env.episode_stats()).to(dev)  # D2H of 8 numbers: syncs this rank's stream
    stats = reduce_episode_stats(stats, force=bool(force))
    sync()
    stats_reduce_us = (time.perf_counter() - ts) * 1e6

    # ---- who ran: one all-gather of (rank, device, PCI bus id, this rank's own median block)
    mine = dict(device_identity(torch, local_rank), rank=rank, lanes=[lane0, lane1], env_steps=(lane1 - lane0) * K,
                block_ms_median=sorted(own_blocks[r] for r in order)[len(order) // 2] * 1e3, launch_period_steady_ms=steady_ms)
    ranks = gather_ranks(mine)
    # an N-GPU line is evidence of N GPUs: every rank on a device of its own, or the job fails (two ranks on one device would
    # still print a line -- at half the speed, or worse, looking like a scaling problem)
    rccl_world = dist.get_world_size() if dist is not None else 1
    if world > 1:
        ids = [(r.get("pci_bus_id"), r.get("uuid")) for r in ranks]
        if len(set(ids)) != world or rccl_world != world:
            sys.stderr.write("bench.py: %d ranks on %d distinct devices (process group of %d): %r\n" % (world, len(set(ids)), rccl_world, ids))
            sys.exit(4)

    # ---- each launch bracketed by its own event pair (the events themselves add about 2 us, so this reads high)
    k = min(args.kernel_timing_launches, max(1, args.steps))
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(k)]
    for t in range(k):
        starts[t].record()
        env.step_tensors(actions[t % ring])
        stops[t].record()
    torch.cuda.synchronize()
    per = sorted(starts[t].elapsed_time(stops[t]) for t in range(k))
    kern_ms = float(np.mean(per))
    kern_ms_med = per[len(per) // 2]
    env.check_actions()
    fused = None
    if args.rollout_k > 0:
        KF = min(args.rollout_k, ring)
        ro_acts = actions[:KF].contiguous()
        ro_out = (torch.empty((KF, n, env.obs_dim), dtype=env.obs_dtype, device=dev),
                  torch.empty((KF, n), dtype=torch.int32, device=dev),
                  torch.empty((KF, n), dtype=torch.uint8, device=dev), torch.empty((KF, n), dtype=torch.uint8, device=dev))
        launches = max(8, args.steps // KF)
        for _ in range(2):
            env.rollout(ro_acts, out=ro_out)
        barrier()
        tf = time.perf_counter()
        for _ in range(launches):
            env.rollout(ro_acts, out=ro_out)
        barrier()
        dtf = time.perf_counter() - tf
        if dist is not None:
            tt = torch.tensor([dtf], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtf = float(tt.item())
        fused = {"value": float(n) * KF * launches * world / dtf, "unit": "env-steps/s", "ticks_per_launch": KF,
                 "launches": launches, "us_per_tick": dtf / (KF * launches) * 1e6,
                 "note": "sf_rollout: K ticks fused into one launch (state stays in registers), actions of all K "
                         "ticks resident up front, obs/reward/done/info written for every tick; bit-identical to K "
                         "sf_step launches (tests/test_gpu_parity.py::test_fused_rollout_equals_single_steps)"}
        del ro_out
    solo = rank == 0 and world == 1  # the single-GPU extras: not part of an N > 1 run (its ranks would wait for rank 0)
    # a measured ceiling beside the 8 TB/s spec peak: device-to-device copy of 1 GiB, read + write bytes
    copy_gbs = None
    if solo:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 2.0 * src.numel() * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst
    image_obs = None
    if args.image_envs > 0 and solo:
        ni = args.image_envs
        from spacefortress_amd import FrameStack
        ienv = SFVecEnv(ni, gametype=args.gametype, obs_type="image", device=dev, spawn_stride=1, reuse_buffers=True)
        stack = FrameStack(ienv, 4)  # BASELINE configs[4]: 84x84 grey raster + 4-frame stack
        stack.reset()
        iacts = actions[:, :ni].contiguous() if ni <= n else torch.randint(0, ienv.n_actions, (ring, ni), device=dev,
                                                                           dtype=torch.uint8, generator=g)
        isteps = max(200, min(1000, args.steps // 4))
        for t in range(400):  # into mid-episode states: missiles, shells, explosions on screen
            stack.step(iacts[t % ring])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(isteps):
            stack.step(iacts[t % ring])
        e1.record()
        torch.cuda.synchronize()
        ims = e0.elapsed_time(e1) / isteps
        floor_us = (ni * (84 * 84 + 1200)) / 6.3e12 * 1e6  # 7 056 B written + about 1.2 KB of state read per env at 6.3 TB/s
        i_ach = IMAGE_ALGO_BYTES * ni / (ims * 1e-3) / 1e9
        iprof = committed_image_profile(ni)
        ienv.check_state()
        image_obs = {"state_ok": True,
                     "kernel": "sf_render_kernel<true>", "kernel_ms_rocprof": iprof.get("kernel_ms_rocprof"),
                     "kernel_ms_rocprof_median": iprof.get("kernel_ms_rocprof_median"),
                     "kernel_ms_rocprof_stale": iprof.get("stale") if iprof else None,
                     "kernel_ms_rocprof_source": ("rocprofv3 --kernel-trace mean of build %s (profiles/%s), replayed"
                                                  % (iprof.get("sf_build_id"), iprof.get("stats_file"))) if iprof else None,
                     "value": ni / ims * 1e3, "unit": "env-steps/s", "envs": ni, "steps": isteps, "us_per_step": ims * 1e3,
                     "frame_bytes_per_step": ni * 84 * 84, "frames_GBps": ni * 84 * 84 / ims / 1e6,
                     "output_floor_us": floor_us, "frac_of_output_floor": floor_us / (ims * 1e3),
                     "roofline": {"bound": "hbm", "achieved": i_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": i_ach / HBM_PEAK_GBS, "algorithmic_bytes_per_env_step": IMAGE_ALGO_BYTES,
                                  "algorithmic_bytes_per_launch": IMAGE_ALGO_BYTES * ni,
                                  "traffic": iprof.get("traffic_bytes_per_launch"),
                                  "traffic_source": ("HBM-side bytes per launch of sf_render_kernel<true> from the committed --pmc passes of build %s "
                                                     "(profiles/image_kernel_latest.json: FETCH_SIZE x 2 + WRITE_SIZE), replayed%s"
                                                     % (iprof.get("sf_build_id"), " -- STALE: another build is loaded" if iprof.get("stale") else "")) if iprof.get("traffic_bytes_per_launch") else None,
                                  "note": "SURVEY 8(d)'s 7 448 B per env-step over the whole step (sf_step + sf_render_stack); "
                                          "the render kernel restates cairo's scan converter exactly and is latency- / issue-bound, not memory-bound (DESIGN.md 5)"},
                     "note": "BASELINE configs[4]: youturn image obs, 84x84 grey raster + 4-frame stack (device ring "
                             "[N,4,84,84], one new frame per env and step, finished envs' older slots zeroed by the same launch), "
                             "sf_step + sf_render_stack per step; one wave per env rasterises the 90x92 frame in LDS (INTER_AREA to 84x84); HIP "
                             "events; every frame equals the reference's own cairo 1.16 renderer bit for bit (tests/golden/frames, DESIGN.md 5)"}
        ienv.close()
    configs = None
    if solo and not args.no_configs:
        # every other configuration of BASELINE.json, each a few graph-replayed blocks on this GPU (they are parity-test
        # sizes, not the metric: `value` above stays on the metric's configuration)
        configs = []
        cases = [("configs[1]: youturn, 4096 envs, features obs", "youturn", 4096),
                 ("configs[2]: autoturn, 65536 envs (reduced action set)", "autoturn", 65536),
                 ("configs[3]: youturn, 262144 envs over 8 GPUs -- one GPU's share, 32768 envs, per-lane auto-reset", "youturn", 32768),
                 ("youturn, 262144 envs on ONE GPU (the largest single-GPU batch: no longer Infinity-Cache resident)", "youturn", 262144)]
        KC, reps = 256, 7
        for name, gt, ne in cases:
            if gt == args.gametype and ne == n:
                continue
            cenv = SFVecEnv(ne, gametype=gt, obs_type="features", device=dev, spawn_stride=1, reuse_buffers=True)
            cacts = torch.randint(0, cenv.n_actions, (ring, ne), device=dev, dtype=torch.uint8, generator=g)
            crows = [cacts[k] for k in range(ring)]
            cenv.reset()
            for t in range(64):
                cenv.step_tensors(crows[t % ring])
            cg = capture(lambda k: cenv.step_tensors(crows[k % ring]), KC)
            ms = []
            for _ in range(reps):
                ev0.record()
                cg.replay()
                ev1.record()
                sync()
                ms.append(ev0.elapsed_time(ev1) / KC)
            ms.sort()
            m = ms[len(ms) // 2]
            ach = ALGO_BYTES[gt] * ne / (m * 1e-3) / 1e9
            configs.append({"name": name, "gametype": gt, "envs": ne, "ms_per_step": m, "value": ne / m * 1e3,
                            "unit": "env-steps/s", "launches_per_block": KC, "blocks": reps,
                            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ALGO_BYTES[gt] * ne}})
            cenv.check_actions()
            del cg
            cenv.close()
        # the reference's observation dtype: _get_features returns float64 (ENV:134-157); the headline times float32 rows
        # (SURVEY a14 / 8d).  Same batch, same kernel family, 8 bytes per feature: 464 + 4 * 19 algorithmic bytes per env-step
        if True:
            cenv = SFVecEnv(n, gametype=args.gametype, obs_type="features", device=dev, spawn_stride=1, reuse_buffers=True,
                            obs_dtype=torch.float64)
            cacts = torch.randint(0, cenv.n_actions, (ring, n), device=dev, dtype=torch.uint8, generator=g)
            crows = [cacts[k] for k in range(ring)]
            cenv.reset()
            for t in range(64):
                cenv.step_tensors(crows[t % ring])
            cg = capture(lambda k: cenv.step_tensors(crows[k % ring]), KC)
            ms = []
            for _ in range(reps):
                ev0.record()
                cg.replay()
                ev1.record()
                sync()
                ms.append(ev0.elapsed_time(ev1) / KC)
            ms.sort()
            m = ms[len(ms) // 2]
            b64 = ALGO_BYTES[args.gametype] + 4 * cenv.obs_dim
            ach = b64 * n / (m * 1e-3) / 1e9
            configs.append({"name": "%s, %d envs, float64 observations (SF_FLAG_OBS_F64: the reference's dtype, ENV:134-157)" % (args.gametype, n),
                            "gametype": args.gametype, "envs": n, "obs_dtype": "float64", "ms_per_step": m, "value": n / m * 1e3,
                            "unit": "env-steps/s", "launches_per_block": KC, "blocks": reps,
                            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                         "algorithmic_bytes_per_env_step": b64, "algorithmic_bytes_per_launch": b64 * n,
                                         "note": "SURVEY 8(d)'s B with 8 * D observation bytes instead of 4 * D"}})
            del cg
            cenv.close()
        if image_obs is not None:
            configs.append({"name": "configs[4]: youturn image obs, 84x84 grey raster + 4-frame stack, %d envs" % image_obs["envs"],
                            "gametype": args.gametype, "envs": image_obs["envs"], "ms_per_step": image_obs["us_per_step"] * 1e-3,
                            "value": image_obs["value"], "unit": "env-steps/s", "roofline": image_obs["roofline"]})
    if configs is not None and base and isinstance(base.get("config0"), dict) and "value" in base["config0"]:
        configs.insert(0, base["config0"])
    # (last of the extras: its per-step synchronise leaves the GPU idle most of the time and the clocks drop -- whatever ran
    #  right behind it measured the ramp-up, e.g. a 1 GiB copy at 0.7 TB/s)
    host_api = None
    if args.numpy_api > 0 and solo:
        np_actions = actions.cpu().numpy().astype(np.int64)  # what rl/train.py:79 hands over
        env.step(np_actions[0])
        th = time.perf_counter()
        for t in range(args.numpy_api):
            env.step(np_actions[t % ring])
        dt = time.perf_counter() - th
        host_api = {"value": n * args.numpy_api / dt, "unit": "env-steps/s", "ms_per_step": dt / args.numpy_api * 1e3,
                    "steps": args.numpy_api,
                    "note": "envs.step(cpu_actions) as rl/train.py:79-80 calls it: numpy int64 actions in, numpy "
                            "obs/reward/done/info out every step (PCIe both ways, a synchronise per step)"}
    if os.environ.get("SF_PMC_CALIB"):
        # known-byte calibration dispatches for the rocprofv3 --pmc passes (tools/pmc_report.py):
        # sf_group_copy_kernel reads n*20*16 bytes in the step kernel's own access pattern
        # (16 bytes per lane, 64-lane rows) and writes the same number linearly
        import ctypes
        from spacefortress_amd import _lib
        for which in (0, 1, 0, 1):
            nb = ctypes.c_size_t()
            _lib.check(_lib.lib().sf_calibration_copy(env._h, which, ctypes.byref(nb)))

    env.check_state()
    state_ok = True  # (check_state raises otherwise)
    if rank == 0:
        total_steps = float(n) * K * world
        # the whole job's throughput: the env-steps EVERY rank reports for a timed block (gathered above) over the block's
        # time, MAX over ranks -- i.e. the sum of the ranks' shares at the job's clock (`ranks[*].value_share`)
        value = whole_job_value(ranks, elapsed)
        assert abs(value - total_steps / elapsed) <= 1e-9 * value, (value, total_steps / elapsed)
        algo = ALGO_BYTES[args.gametype] * n
        # `achieved` / `frac`: the algorithmic bytes of one launch over ms_per_step of the blocks that give `value` (the same
        # clock as `value`); the HIP-event launch period of the event-carrying blocks stands beside it as a labelled extra
        achieved = algo / (elapsed / K) / 1e9
        achieved_event = algo / (region_ms * 1e-3) / 1e9
        prof = committed_profile(args.gametype, n, args.obs_type)
        out = {
            "metric": METRIC,
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "repeats": repeats,
            "launch": "hip_graph (K sf_step launches captured once, one replay per block)" if use_graph else "loop (one Python call per launch)",
            "loop_issue": loop_issue,
            "block_ms": {"median": elapsed * 1e3, "min": blocks[order[0]] * 1e3, "max": blocks[order[-1]] * 1e3,
                         "first": blocks[0] * 1e3, "blocks_without_events": len(order),
                         "with_events_median": ev_blocks[len(ev_blocks) // 2] * 1e3 if ev_blocks else None,
                         "blocks_with_events": len(ev_blocks),
                         "note": "each block = exactly `steps` launches between barrier + synchronize brackets (max over "
                                 "ranks).  Every other block also carries the pair of HIP events that measures "
                                 "roofline.launch_period_ms; the two event packets cost such a block about 7 us of its own "
                                 "(tools/block_probe.py), so value and ms_per_step come from the median of the blocks WITHOUT "
                                 "them, and the event blocks' median stands beside it"},
            "value_with_action_gen": total_steps / s_blocks[s_med],
            "action_gen": {"value": total_steps / s_blocks[s_med], "unit": "env-steps/s", "ms_per_step": s_blocks[s_med] / K * 1e3,
                           "launch_period_ms": s_period,
                           "note": "the same timed blocks through sf_step_sampled: every lane draws its action inside the "
                                   "launch (Philox4x32-10 keyed by seed, counter (lane of the job, tick); the tick counter "
                                   "lives on the device, so every graph replay plays new actions); no action tensor is "
                                   "generated, stored or loaded (SURVEY 8d: 'action generation on device ... report both'; "
                                   "tests/test_gpu_sampled.py replays the sampled actions through the oracle)"},
            "stats_reduce_us": stats_reduce_us,
            "config": {"workload": "%s, %d envs/GPU, %s obs (f32), uniform random discrete actions resident in HBM, "
                                   "per-lane auto-reset" % (args.gametype, n, args.obs_type),
                       "envs_per_gpu": n, "gametype": args.gametype, "obs_type": args.obs_type,
                       "parallelism": "%d contiguous lane shard(s) of one %d-lane batch, one process per GPU, no data-path "
                                      "collective" % (world, n * world)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_source": "algorithmic_bytes_per_launch / ms_per_step (the blocks that give `value`)",
                         "achieved_event_period": achieved_event, "frac_event_period": achieved_event / HBM_PEAK_GBS,
                         "event_period_note": "the same bytes over launch_period_ms: HIP events around the launches of the "
                                              "event-carrying blocks (end of the first launch to end of the last)",
                         "frac_note": "of the 8 TB/s HBM spec peak; the 77 MB state of this workload is Infinity-Cache "
                                      "resident (256 MB), so the bytes mostly move between L2 and the Infinity Cache",
                         "measured_copy_ceiling_GBps": copy_gbs,
                         "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None,
                         "traffic": prof.get("traffic_bytes_per_launch"),
                         "stale": prof.get("stale") if prof else None,
                         "profile_build_id": prof.get("sf_build_id") if prof else None,
                         "loaded_build_id": prof.get("loaded_build_id") if prof else None,
                         "stale_note": "traffic and kernel_ms_rocprof are replayed from the committed rocprofv3 runs of the "
                                       "build named in profile_build_id; stale = the loaded library is another build",
                         "traffic_source": ("replayed from the committed rocprofv3 --pmc passes of kernel version %s "
                                            "(profiles/%s), not measured by this run" % (prof.get("version"), prof.get("pmc_file")))
                                           if prof else None,
                         "kernel": "sf_step_kernel", "launch_period_ms": region_ms,
                         "launch_period_steady_ms": steady_ms, "frac_steady": algo / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "steady_note": "%d back-to-back launches (%.1f s of the Python loop) outside the timed blocks; `achieved` / "
                                        "`frac` use the timed blocks' ms_per_step, which for a block of a few launches includes its "
                                        "start on an idle GPU" % (n_steady, steady_total_ms * 1e-3),
                         "kernel_ms_rocprof": prof.get("kernel_ms_rocprof"),
                         "kernel_ms_rocprof_median": prof.get("kernel_ms_rocprof_median"),
                         "kernel_ms_rocprof_source": ("rocprofv3 --kernel-trace mean of kernel version %s (profiles/%s)"
                                                      % (prof.get("version"), prof.get("trace_file"))) if prof else None,
                         "kernel_ms_event_pair_mean": kern_ms, "kernel_ms_event_pair_median": kern_ms_med,
                         "algorithmic_bytes_per_launch": algo, "launches_timed": K},
            "cpu_baseline": base,
            "state_ok": state_ok,
            "state_ok_note": "env.check_state() (sf_check_state: the sticky device-side words -- a split launch's hand-over that timed "
                             "out, a per-episode counter or key timer that left its packed width) after the timed blocks, after the steady "
                             "region and at the end: it raises on any error, so a line that prints has passed all three",
            "ranks": ranks,
            "rccl_world": rccl_world,
            "rollout_fused": fused,
            "host_api": host_api,
            "image_obs": image_obs,
            "configs": configs,
            "episode_stats": summarize(stats.cpu()),
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
