"""Timing of the general renderer (sf_render_generic.hip: any geometry SSF_Env(scale, viewport, ls) names) beside the default
geometry's frame kernel.   python tools/geometry_probe.py [n_envs] [steps]   (GPU box)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import SFVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
modes = sys.argv[3:] or ["image"]
for name, geom in (("default .2 (130,80,450,460) ls 3 -> 90x92", None), (".25 (100,60,500,520) ls 2 -> 125x130", (.25, (100, 60, 500, 520), 2)),
                   (".4 (130,80,450,460) ls 3 -> 180x184", (.4, (130, 80, 450, 460), 3))):
  for mode in modes:
    env = SFVecEnv(n, gametype="youturn", obs_type=mode, spawn_stride=1, reuse_buffers=True, image_geometry=geom)
    env.reset()
    acts = torch.randint(0, env.n_actions, (64, n), device=env.device, dtype=torch.uint8)
    for t in range(300):
        env.step_tensors(acts[t % 64])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps):
        env.step_tensors(acts[t % 64])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    print("%-44s n=%d: step + %s frame %.1f us per step (%.3g env-steps/s)" % (name, n, mode, us, n / us * 1e6), flush=True)
    env.close()
