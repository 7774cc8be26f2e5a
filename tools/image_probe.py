"""Timing of the image observation: sf_step + sf_render per step, and sf_render alone (HIP events).
    python tools/image_probe.py [n_envs] [steps]"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from spacefortress_amd import SFVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
args = sys.argv[3:]
hunter = "hunter" in args  # a firing pattern that destroys the fortress every few seconds (tools/soak.py), autoturn game
args = [a for a in args if a != "hunter"]
for mode in ([a for a in args if a != "stack"] or ([] if args else ["image", "image-raw"])):
    env = SFVecEnv(n, gametype="autoturn" if hunter else "youturn", obs_type=mode, spawn_stride=1, reuse_buffers=True)
    env.reset()
    acts = torch.randint(0, env.n_actions, (64, n), device=env.device, dtype=torch.uint8)
    if hunter:
        rng = np.random.default_rng(1)
        pat = np.array(([1] + [0] * 7) * 11 + [1, 0, 1, 0] + [0] * 4, np.uint8)
        assert len(pat) == 96
        ph = rng.integers(0, 96, n)
        rows = np.stack([np.where(rng.random(n) < 0.1, rng.integers(0, env.n_actions, n), pat[(t + ph) % 96]) for t in range(96)])
        acts = torch.from_numpy(rows.astype(np.uint8)).to(env.device)
    ring = acts.shape[0]
    for t in range(400):  # get into mid-episode states (missiles, shells, explosions)
        env.step_tensors(acts[t % ring])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps):
        env.step_tensors(acts[t % ring])
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
    # ... and the step launch of an image batch alone (no frame): sf_step with obs_dev = NULL
    import ctypes as C
    from spacefortress_amd import _lib
    _, rew, done, info = env._alloc()
    e0.record()
    for t in range(steps):
        a = acts[t % ring]
        _lib.check(env._L.sf_step(env._h, C.c_void_p(a.data_ptr()), 1, None, C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
                                  C.c_void_p(info.data_ptr()), env._stream()))
    e1.record()
    torch.cuda.synchronize()
    ms_s = e0.elapsed_time(e1) / steps
    print(("hunter " if hunter else "") + "%s n=%d: step+render %.1f us (%.3g frames/s), render alone %.1f us, output %.2f GB/s, step alone %.2f us" %
          (mode, n, ms * 1e3, n / ms * 1e3, ms_r * 1e3, out.numel() / ms_r / 1e6, ms_s * 1e3))
    env.close()

if "stack" in sys.argv[3:] or not sys.argv[3:]:
    # BASELINE configs[4] as bench.py times it: the 4-frame ring, sf_step + sf_render_stack per step
    from spacefortress_amd import FrameStack
    env = SFVecEnv(n, gametype="youturn", obs_type="image", spawn_stride=1, reuse_buffers=True)
    st = FrameStack(env, 4)
    st.reset()
    acts = torch.randint(0, env.n_actions, (64, n), device=env.device, dtype=torch.uint8)
    for t in range(400):
        st.step(acts[t % 64])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps):
        st.step(acts[t % 64])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print("stack n=%d: step+render into the 4-frame ring %.1f us (%.3g frames/s)" % (n, ms * 1e3, n / ms * 1e3))
    env.close()
