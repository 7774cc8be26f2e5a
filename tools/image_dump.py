"""Debug aid: replay a golden on the GPU and save raw frames at the given ticks to gpurun_out/."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spacefortress_amd import SFVecEnv
name = sys.argv[1]
ticks = [int(t) for t in sys.argv[2:]]
z = np.load(os.path.join(ROOT, "tests/golden", name + ".npz"))
meta = json.loads(str(z["meta"]))
env = SFVecEnv(1, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"],
               spawn_skip=meta["spawn_skip"], obs_type="image-raw")
# no reset(): the recording starts from the first Game (sf_create)
out = {}
for t in range(max(ticks) + 1):
    obs, *_ = env.step_tensors(torch.tensor([int(z["actions"][t])], dtype=torch.uint8, device=env.device))
    if t in ticks:
        out["t%d" % t] = obs[0].cpu().numpy()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "frames_%s.npz" % name), **out)
print("saved", list(out))
