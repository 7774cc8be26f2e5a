"""Where the time of the numpy-in / numpy-out step goes (65 536 envs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import SFVecEnv
n = 65536
env = SFVecEnv(n, spawn_stride=1, reuse_buffers=True)
env.reset()
acts = np.random.randint(0, 5, (64, n)).astype(np.int64)
def t(fn, k=100):
    fn(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(k): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("step(numpy) as shipped:        %.3f ms" % t(lambda i: env.step(acts[i % 64])))
dev = env.device
a_pin = torch.empty(n, dtype=torch.int64).pin_memory()
a_dev = torch.empty(n, dtype=torch.int64, device=dev)
obs_pin = torch.empty((n, 19), dtype=torch.float32).pin_memory()
rew_pin = torch.empty(n, dtype=torch.int32).pin_memory()
d_pin = torch.empty(n, dtype=torch.uint8).pin_memory(); i_pin = torch.empty(n, dtype=torch.uint8).pin_memory()
def pinned(i, copy_out):
    a_pin.numpy()[:] = acts[i % 64]
    a_dev.copy_(a_pin, non_blocking=True)
    o, r, d, inf = env.step_tensors(a_dev)
    obs_pin.copy_(o, non_blocking=True); rew_pin.copy_(r, non_blocking=True); d_pin.copy_(d, non_blocking=True); i_pin.copy_(inf, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    if copy_out:
        return obs_pin.numpy().copy(), rew_pin.numpy().astype(np.int64), d_pin.numpy().astype(bool), i_pin.numpy().astype(bool)
    return obs_pin.numpy(), rew_pin.numpy(), d_pin.numpy(), i_pin.numpy()
print("pinned staging, views returned: %.3f ms" % t(lambda i: pinned(i, False)))
print("pinned staging, fresh copies:   %.3f ms" % t(lambda i: pinned(i, True)))
