"""GPU debug: one fixture frame by label against the reference's frame: where they differ."""
import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import spacefortress_amd as sfa
from sfcompare import snapshots_to_fields
name, label = sys.argv[1], sys.argv[2]
z = np.load("tests/golden/frames/" + name)
i = [str(x) for x in z["labels"]].index(label)
snap = z["snaps"][i:i + 1]
env = sfa.SFVecEnv(1, gametype="youturn", obs_type="image-raw")
for k, v in snapshots_to_fields(snap).items(): env.set_field(k, v)
got = env.render("image-raw").cpu().numpy()[0]
want = z["frames"][i]
d = np.abs(got[9:].astype(int) - want[9:].astype(int))
ys, xs = np.nonzero(d)
cx, cy = (snap["ship_x"][0] - 130) * .2, (snap["ship_y"][0] - 80) * .2
print("centre", cx, cy, "n", len(ys), "max", d.max() if len(ys) else 0)
for a, b in list(zip(ys + 9, xs))[:40]:
    print(a, b, "r", np.hypot(b + .5 - cx, a + .5 - cy) / .2, "got", got[a, b], "want", want[a, b])
