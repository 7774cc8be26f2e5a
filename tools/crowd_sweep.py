"""More seeds of tests/test_gpu_image.py::test_crowded_constructed_frames_vs_model: constructed frames with up to 20 missiles and 20
shells anywhere, both sizes against the numpy model (round 3: 12 seeds x 96 frames x 2 sizes = 2 304 compared, 0 mismatches)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spacefortress_amd as sfa
from oracle import oracle as O
from oracle import render_np as R
from test_gpu_parity import _fuzz_base, _load_both
z = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))
hb, hs = z["hex_points"][:12], z["hex_points"][12:]
bg = R.background(hb, hs)
tot = bad = 0
for gt in ("youturn", "autoturn"):
    for seed in range(100, 106):
        n = 96
        base, pv = _fuzz_base(O, gt, n, np.random.default_rng(seed))
        ang = np.degrees(np.arctan2(base["shell_vy"], base["shell_vx"]))
        base["shell_angle"] = np.where(ang < 0, ang + 360.0, ang)
        env, orc = _load_both(sfa, O, gt, base, prev_vlner=pv)
        raw = env.render("image-raw").cpu().numpy(); small = env.render("image").cpu().numpy()
        snaps = orc.snapshots()
        for i in range(n):
            want = R.render_raw(snaps[i], hb, hs, bg=bg)
            for got, w in ((raw[i], want), (small[i, 0], R.resize_area(want))):
                d = np.abs(got.astype(int) - w.astype(int)); tot += 1
                if d.max() > 2 or (d == 0).mean() < 0.995:
                    bad += 1; print("MISMATCH", gt, seed, i, got.shape, int(d.max()))
        env.close()
print("crowded constructed frames: %d compared, %d mismatches" % (tot, bad))
