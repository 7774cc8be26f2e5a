"""GPU debug: the general renderer's frames against the model at one geometry (explosions at many places)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import spacefortress_amd as sfa
from oracle import render_np as R, oracle as O
from sfcompare import snapshots_to_fields
scale = float(sys.argv[1]) if len(sys.argv) > 1 else .4
vp, ls = (130, 80, 450, 460), 3
z = np.load("tests/golden/tables.npz"); hb, hs = z["hex_points"][:12], z["hex_points"][12:]
g = np.load("tests/golden/frames/geometries.npz")
base = g["snaps"][5].copy()
rng = np.random.default_rng(1)
N = 64
snaps = np.array([base] * N, O.SNAPSHOT_DTYPE)
for i in range(N):
    snaps[i]["ship_alive"] = 0 if i % 2 else 1
    snaps[i]["ship_x"], snaps[i]["ship_y"] = rng.uniform(200, 500), rng.uniform(150, 480)
    snaps[i]["fort_alive"] = i % 3 == 0
    snaps[i]["fort_angle"] = 10 * (i % 36)
env = sfa.SFVecEnv(N, gametype="youturn", obs_type="image-raw", image_geometry=(scale, vp, ls))
for k, v in snapshots_to_fields(snaps).items(): env.set_field(k, v)
got = env.render("image-raw").cpu().numpy()
R.set_geometry(scale, vp, ls)
bad = 0
for i in range(N):
    want = R.render_raw(snaps[i], hb, hs)
    if not np.array_equal(got[i], want):
        bad += 1
        d = np.abs(got[i].astype(int) - want)
        ys, xs = np.nonzero(d)
        print("lane", i, "ship_alive", snaps[i]["ship_alive"], "fort_alive", snaps[i]["fort_alive"], "max", d.max(), "n", len(ys),
              [(int(a), int(b), int(got[i][a, b]), int(want[a, b])) for a, b in list(zip(ys, xs))[:6]],
              "ship at", (snaps[i]["ship_x"] - 130) * R.SX, (snaps[i]["ship_y"] - 80) * R.SY)
print("bad", bad, "of", N)
