"""What a K-launch timed block of bench.py pays beyond its K launches: the HIP events around the replay, the way the host waits.
    python tools/block_probe.py [K] [blocks]   (GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import SFVecEnv

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 65536
dev = torch.device("cuda", 0)
env = SFVecEnv(n, gametype="youturn", device=dev, spawn_stride=1, reuse_buffers=True)
acts = torch.randint(0, 5, (64, n), device=dev, dtype=torch.uint8)
rows = [acts[k] for k in range(64)]
env.reset()
for t in range(50):
    env.step_tensors(rows[t % 64])
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for k in range(K):
            env.step_tensors(rows[k % 64])
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream(dev)


def blocks(mode):
    out = []
    for _ in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "events+query":
            ev0.record(); g.replay(); ev1.record()
            while not ev1.query():
                pass
        elif mode == "stream.query":
            g.replay()
            while not st.query():
                pass
        elif mode == "event_after+query":
            g.replay(); ev1.record()
            while not ev1.query():
                pass
        else:
            g.replay()
        torch.cuda.synchronize()
        out.append(time.perf_counter() - t0)
    out.sort()
    return out[len(out) // 2] * 1e6, out[0] * 1e6


for rnd in range(2):
    for mode in ("events+query", "event_after+query", "stream.query", "sync only"):
        med, mn = blocks(mode)
        print("%-18s K=%d: median block %.1f us (%.2f us per step, %.3g env-steps/s), min %.1f" % (mode, K, med, med / K, n * K / med * 1e6, mn), flush=True)
env.close()
