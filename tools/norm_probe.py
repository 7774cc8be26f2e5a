import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import SFVecEnv, SFVecNormalize
n = 65536
vn = SFVecNormalize(SFVecEnv(n, spawn_stride=1, reuse_buffers=True))
obs = vn.reset()
rew = torch.zeros(n, dtype=torch.int32, device=obs.device)
for _ in range(50): vn._filter(obs, rew)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(1000): vn._filter(obs, rew)
e1.record(); torch.cuda.synchronize()
print("normalize (reduce + apply): %.2f us" % (e0.elapsed_time(e1)))
