"""Timing of the image observation: sf_step + sf_render per step, and sf_render alone (HIP events).
    python tools/image_probe.py [n_envs] [steps]"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from spacefortress_amd import SFVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for mode in (sys.argv[3:] or ["image", "image-raw"]):
    env = SFVecEnv(n, gametype="youturn", obs_type=mode, spawn_stride=1, reuse_buffers=True)
    env.reset()
    acts = torch.randint(0, 5, (64, n), device=env.device, dtype=torch.uint8)
    for t in range(400):  # get into mid-episode states (missiles, shells, explosions)
        env.step_tensors(acts[t % 64])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps):
        env.step_tensors(acts[t % 64])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    out = torch.empty((n,) + env.obs_shape, dtype=torch.uint8, device=env.device)
    e0.record()
    for t in range(steps):
        env.render(mode, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms_r = e0.elapsed_time(e1) / steps
    print("%s n=%d: step+render %.1f us (%.3g frames/s), render alone %.1f us, output %.2f GB/s" %
          (mode, n, ms * 1e3, n / ms * 1e3, ms_r * 1e3, out.numel() / ms_r / 1e6))
    env.close()
