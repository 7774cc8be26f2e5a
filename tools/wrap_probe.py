"""Per-step cost of the wrappers around sf_step at 65 536 envs (HIP events): plain, +VecNormalize,
DeviceRollout.step (+ record), compute_returns."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import SFVecEnv, SFVecNormalize, DeviceRollout

n, K = 65536, 2000
dev = torch.device("cuda")
acts = torch.randint(0, 5, (64, n), device=dev, dtype=torch.uint8)


def timed(fn, k=K):
    for t in range(100):
        fn(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(k):
        fn(t)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


env = SFVecEnv(n, spawn_stride=1, reuse_buffers=True)
env.reset()
print("sf_step                      %.2f us" % timed(lambda t: env.step_tensors(acts[t % 64])))
vn = SFVecNormalize(SFVecEnv(n, spawn_stride=1, reuse_buffers=True))
vn.reset()
print("sf_step + VecNormalize       %.2f us" % timed(lambda t: vn.step_tensors(acts[t % 64])))
T = 20
ro = DeviceRollout(SFVecEnv(n, spawn_stride=1), T)
ro.reset()
print("DeviceRollout.step           %.2f us" % timed(lambda t: ro.step(t % T, acts[t % 64])))
nv = torch.zeros(n, 1, device=dev)
print("compute_returns (T=20)       %.2f us" % timed(lambda t: ro.compute_returns(nv, True, 0.99, 0.95), 200))
