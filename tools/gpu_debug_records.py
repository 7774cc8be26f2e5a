import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spacefortress_amd as sfa
N = 64
env = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=1)
env.reset()
rng = np.random.default_rng(0)
for t in range(12):
    a = torch.from_numpy(rng.integers(0, 5, N).astype(np.uint8)).to(env.device)
    env.step_tensors(a)
    ra, rb = env.draw_records(False), env.draw_records(True)
    objmask = ra[:, 12:16].copy().view(np.uint32)[:, 0]
    ha, hb = ra[:, 384:424].copy().view(np.int16), rb[:, 384:424].copy().view(np.int16)
    for i in range(N):
        for s in range(20):
            if (objmask[i] >> (2 + s)) & 1 and ha[i, s] != hb[i, s]:
                print("t", t, "lane", i, "slot", s, "step", ha[i, s], "state", hb[i, s], "ship angle", env.get_field("ship_angle")[i])
print("done")
