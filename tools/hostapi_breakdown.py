"""Where a host-API step's time goes (envs.step(numpy actions) -> numpy results, rl/train.py:79-80), 65 536 envs."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from spacefortress_amd import SFVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = SFVecEnv(n, gametype="youturn")
env.reset()
rng = np.random.default_rng(0)
acts = rng.integers(0, env.n_actions, n).astype(np.int64)
def t(f, k=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("step(numpy) whole                 %.3f ms" % t(lambda: env.step(acts)))
dev = env.device
print("actions: range check (min/max)     %.3f ms" % t(lambda: (acts.min(), acts.max())))
print("actions: from_numpy().to(device)   %.3f ms" % t(lambda: torch.from_numpy(acts).to(dev)))
a_dev = torch.from_numpy(acts).to(dev)
print("step_tensors                       %.3f ms" % t(lambda: env.step_tensors(a_dev)))
o, r, d, i = env.step_tensors(a_dev)
print("obs %s %s" % (tuple(o.shape), o.dtype))
print("obs.cpu().numpy()                  %.3f ms" % t(lambda: o.cpu().numpy()))
print("rew.cpu().numpy().astype(int64)    %.3f ms" % t(lambda: r.cpu().numpy().astype(np.int64)))
print("done + info .cpu().numpy().astype  %.3f ms" % t(lambda: (d.cpu().numpy().astype(bool), i.cpu().numpy().astype(bool))))
ho = torch.empty(o.shape, dtype=o.dtype, pin_memory=True)
def pinned():
    ho.copy_(o, non_blocking=True); torch.cuda.current_stream().synchronize()
print("obs -> pinned buffer (async + sync) %.3f ms" % t(pinned))
print("... + numpy copy out of it          %.3f ms" % t(lambda: (pinned(), ho.numpy().copy())))
base = d._base
print("three small copies (rew, done, info)  %.3f ms" % t(lambda: (r.cpu().numpy().astype(np.int64), d.cpu().numpy().astype(bool), i.cpu().numpy().astype(bool))))
if base is not None:
    def one():
        h = base.cpu().numpy()
        return h[:4 * n].view(np.int32).astype(np.int64), h[4 * n:5 * n].astype(bool), h[5 * n:].astype(bool)
    print("one copy of the shared allocation     %.3f ms" % t(one))
print("step(numpy) whole, again              %.3f ms" % t(lambda: env.step(acts)))
