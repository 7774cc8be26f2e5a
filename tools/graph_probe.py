"""Per-step time of K sf_step launches replayed from a HIP graph vs launched one by one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import SFVecEnv
n, K = 65536, 64
env = SFVecEnv(n, spawn_stride=1, reuse_buffers=True)
env.reset()
acts = torch.randint(0, 5, (K, n), device="cuda", dtype=torch.uint8)
for k in range(K): env.step_tensors(acts[k])
torch.cuda.synchronize()
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / K * 1e3
def direct():
    for k in range(K): env.step_tensors(acts[k])
print("direct launches: %.2f us/step" % timed(direct, 30))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        for k in range(K): env.step_tensors(acts[k])
graph.replay(); torch.cuda.synchronize()
print("graph replay:    %.2f us/step" % timed(graph.replay, 30))
