"""Diagnostics: frames of tests/golden/frames/scores.npz (objects under the score text) that differ from the reference's, with
the pixels around the text.  python tools/gpu_debug_score.py [index ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from spacefortress_amd import SFVecEnv
from sfcompare import snapshots_to_fields
z = np.load(os.path.join(ROOT, "tests", "golden", "frames", "scores.npz"))
frames, snaps = z["frames"], z["snaps"]
env = SFVecEnv(len(snaps), gametype="youturn", obs_type="image-raw")
for k, v in snapshots_to_fields(snaps).items():
    env.set_field(k, v)
got = env.render("image-raw").cpu().numpy()
bad = [i for i in range(len(snaps)) if not np.array_equal(got[i], frames[i])]
print("differ:", bad)
np.set_printoptions(linewidth=250)
rec = env.draw_records(True)
for i in (bad[:4] if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
    s = snaps[i]
    print("frame", i, "points", s["points"], "ship", s["ship_alive"], s["ship_x"], s["ship_y"], s["ship_angle"],
          "missiles", [(float(s["missile_x"][j]), float(s["missile_y"][j]), int(s["missile_angle"][j])) for j in np.flatnonzero(s["missile_alive"])],
          "shells", int(s["shell_alive"].sum()))
    hdr = np.frombuffer(np.asarray(rec[i]).tobytes()[:32], np.uint32)
    print(" header", [hex(int(v)) for v in hdr])
    d = got[i].astype(int) - frames[i].astype(int)
    ys, xs = np.nonzero(d)
    y0, y1, x0, x1 = max(ys.min() - 1, 0), ys.max() + 2, max(xs.min() - 2, 0), xs.max() + 3
    print(" got\n", got[i][y0:y1, x0:x1], "\n want\n", frames[i][y0:y1, x0:x1])
