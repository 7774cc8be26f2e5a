import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from spacefortress_amd import SFVecEnv
n=16384
env=SFVecEnv(n, gametype="youturn", obs_type="image", spawn_stride=1, reuse_buffers=True)
env.reset()
acts=torch.randint(0, env.n_actions, (64,n), device=env.device, dtype=torch.uint8)
for t in range(500): env.step_tensors(acts[t%64])
sd=env.state_dict()
mm=np.array([bin(int(x)).count("1") for x in sd["missile_mask"]]); sm=np.array([bin(int(x)).count("1") for x in sd["shell_mask"]])
alive=(np.asarray(sd["flags"])&1).astype(bool)
print("ship alive %.3f"%alive.mean(), "missiles mean %.2f"%mm.mean(), "hist", np.bincount(mm)[:12], "shells mean %.2f"%sm.mean(), np.bincount(sm)[:6])
quads=3*alive+3*mm+4*sm
print("quads mean %.1f"%quads.mean(), "P(>16) %.3f"%(quads>16).mean(), "calls est", np.mean(np.ceil(np.maximum(quads,1)/16)))
