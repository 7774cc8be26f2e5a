"""CPU-side checks of the image observation: the library's host tables against the numpy model
(oracle/render_np.py), and properties of the model itself.  No GPU."""
import ctypes as C
import os

import numpy as np

from conftest import GOLDEN


def _lib():
    from spacefortress_amd import _lib as L
    return L.lib()


def _hex():
    z = np.load(os.path.join(GOLDEN, "tables.npz"))
    return z["hex_points"][:12], z["hex_points"][12:]  # recorded from the reference's Hexagon::setRadius


def test_background_matches_model_and_reference_hexagons():
    from oracle import render_np as R
    L = _lib()
    bg = np.zeros((92, 90), np.uint8)
    assert L.sf_image_background(bg.ctypes.data_as(C.c_void_p)) == 0
    hb, hs = _hex()
    assert np.array_equal(bg, R.background(hb, hs))
    # the stroke is where the hexagon is: vertices (device space) sit on lit pixels, the centre is dark,
    # total coverage = perimeter * line width (miter joins close the corners exactly)
    for pts in (hb, hs):
        p = (pts.reshape(6, 2) - (130, 80)) * 0.2
        for x, y in p:
            assert bg[int(min(y, 91.9)), int(min(x, 89.9))] > 0 or bg[int(y) - 1, int(x)] > 0
    assert bg[47, 45] == 0
    per = sum(np.hypot(*(q - p)) for pts in (hb, hs) for p, q in zip(pts.reshape(6, 2), np.roll(pts.reshape(6, 2), -1, 0)))
    assert abs(bg.astype(np.float64).sum() / 255.0 - per * 0.2 * 0.6) < 0.02 * per * 0.2 * 0.6
    assert L.sf_image_background(None) < 0


def test_resize_tables_follow_opencv_area():
    from oracle import render_np as R
    L = _lib()
    for ss, ds in ((90, 84), (92, 84)):
        f, c, a = np.zeros(ds, np.int32), np.zeros(ds, np.int32), np.zeros((ds, 4), np.float32)
        assert L.sf_resize_area_tab(ss, ds, f.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                                    a.ctypes.data_as(C.c_void_p)) == 0
        tab = R.area_tab(ss, ds)
        k = 0
        for d in range(ds):
            for j in range(c[d]):
                assert tab[k] == (d, f[d] + j, a[d, j])
                k += 1
            assert (a[d, c[d]:] == 0).all()
        assert k == len(tab) and c.max() <= 3
        assert np.allclose(a.sum(1), 1.0, atol=1e-6)
        assert f[0] == 0 and f[-1] + c[-1] == ss  # covers the source exactly
    i32 = np.zeros(4, np.int32)
    assert L.sf_resize_area_tab(200, 84, i32.ctypes.data_as(C.c_void_p), i32.ctypes.data_as(C.c_void_p),
                                i32.ctypes.data_as(C.c_void_p)) < 0  # scale >= 2: not this table's path


def test_host_resize_matches_model():
    """The library's own INTER_AREA (used for the static background) against the numpy model."""
    from oracle import render_np as R
    L = _lib()
    rng = np.random.default_rng(3)
    bg = np.zeros((92, 90), np.uint8)
    L.sf_image_background(bg.ctypes.data_as(C.c_void_p))
    for img in (bg, rng.integers(0, 256, (92, 90)).astype(np.uint8), np.full((92, 90), 255, np.uint8)):
        out = np.zeros((84, 84), np.uint8)
        assert L.sf_resize_area_u8(img.ctypes.data_as(C.c_void_p), 90, 92, out.ctypes.data_as(C.c_void_p), 84, 84) == 0
        assert np.array_equal(out, R.resize_area(img))


def test_resize_area_properties():
    from oracle import render_np as R
    rng = np.random.default_rng(0)
    flat = np.full((92, 90), 137, np.uint8)
    assert (R.resize_area(flat) == 137).all()
    img = rng.integers(0, 256, (92, 90)).astype(np.uint8)
    out = R.resize_area(img)
    assert out.shape == (84, 84)
    assert abs(out.mean() - img.mean()) < 0.5        # area averaging preserves the mean
    assert out.std() < img.std()                     # ... and smooths
    # separability: a frame constant along x shrinks like its column profile
    prof = rng.integers(0, 256, 92).astype(np.uint8)
    col = np.repeat(prof[:, None], 90, 1)
    o = R.resize_area(col)
    assert (np.abs(o.astype(int) - o[:, :1].astype(int)) <= 1).all()


def test_model_draws_what_the_state_says(oracle_mod):
    """The model on a fresh game: ship wireframe at the spawn, fortress at the centre, bar at
    vulnerability 0, score 0000000; then the bar grows with `vlner` and turns white when kill-ready."""
    from oracle import render_np as R
    hb, hs = _hex()
    bg = R.background(hb, hs)
    env = oracle_mod.OracleEnv("autoturn")
    env.reset()
    s = env.snapshot().copy()
    f = R.render_raw(s, hb, hs, bg=bg).astype(int)
    d = np.abs(f - bg)
    ex, ey = (float(s["ship_x"]) - 130) * .2, (float(s["ship_y"]) - 80) * .2
    box = d[int(ey) - 5:int(ey) + 6, int(ex) - 5:int(ex) + 6]
    assert box.sum() > 255 * 4
    assert d[41:54, 38:54].sum() > 255 * 8          # fortress
    assert (f[89, 25:65] == 84).all()            # empty bar: .33 grey
    assert d[1:6, 32:58].sum() > 128 * 20           # seven zeros
    s["vlner"] = 4
    f4 = R.render_raw(s, hb, hs, bg=bg)
    assert (f4[89, 25:41] == 168).all() and (f4[89, 41:65] == 84).all()
    s["vlner"] = 12
    s["fort_vuln_timer"] = 100
    assert (R.render_raw(s, hb, hs, bg=bg)[89, 25:65] == 255).all()
    s["fort_vuln_timer"] = 250
    assert (R.render_raw(s, hb, hs, bg=bg)[89, 25:65] == 168).all()
    s["points"] = -12.7
    fn = R.render_raw(s, hb, hs, bg=bg).astype(int)
    assert (fn != f).any()


def test_static_variants_match_model():
    """The baked backgrounds the kernel starts from (score 0000000 / empty bar) equal what the model
    draws stroke by stroke on the bare background."""
    from oracle import render_np as R
    L = _lib()
    hb, hs = _hex()
    bg = R.background(hb, hs)
    for v in range(4):
        out = np.zeros((92, 90), np.uint8)
        assert L.sf_image_static(v, out.ctypes.data_as(C.c_void_p)) == 0
        want = bg.copy()
        if v & 1:
            R.score_text(want, 0)
        if v & 2:
            R.over(want, R.rect_poly(255, 522, 455, 532), 84)
        assert np.array_equal(out, want), v
    assert L.sf_image_static(4, bg.ctypes.data_as(C.c_void_p)) < 0


def test_layout_matches_the_references_own_screenshot():
    """Where things are and how grey they are, against measurements of the reference's documentation
    screenshot (rl/imgs/screens.png, rendered by the reference at scale 1; numbers extracted by
    tests/golden/telemetry/make_layout_golden.py).  Positions are compared relative to the big hexagon, in user
    units, within 3.5 (0.7 pixel of the 0.2-scale observation; the extents are thresholded anti-aliased edges, +-1 each); grey levels exactly.  This pins the layout of
    the score text and the bar and the three greys to real reference output; anti-aliasing stays unpinned."""
    import json
    from oracle import render_np as R
    lay = json.load(open(os.path.join(GOLDEN, "telemetry", "screens_layout.json")))
    hb, _ = _hex()
    hx, hy = hb.reshape(6, 2)[:, 0], hb.reshape(6, 2)[:, 1]
    half = R.LINE_W / 2
    hex_left, hex_top, hex_bottom = hx.min() - half, hy.min() - half, hy.max() + half  # outer edge of the stroke
    text_left = R.TXT_X0 + R.TXT_PAD
    text_right = R.TXT_X0 + 6 * R.TXT_ADV + R.TXT_PAD + R.TXT_W
    model = {"text_l": text_left - hex_left, "text_r": text_right - hex_left, "text_t": R.TXT_TOP - hex_top,
             "text_b": R.TXT_TOP + R.TXT_H - hex_top, "bar_l": 255 - hex_left, "bar_r": 455 - hex_left,
             "bar_t": 522 - hex_bottom, "bar_b": 532 - hex_bottom, "hex_w": hx.max() - hx.min() + 2 * half,
             "hex_h": hy.max() - hy.min() + 2 * half}
    for name, p in lay.items():
        shot = {"text_l": p["text_x"][0] - p["hex_x"][0], "text_r": p["text_x"][1] + 1 - p["hex_x"][0],
                "text_t": p["text_y"][0] - p["hex_y"][0], "text_b": p["text_y"][1] + 1 - p["hex_y"][0],
                "bar_l": p["bar_x"][0] - p["hex_x"][0], "bar_r": p["bar_x"][1] + 1 - p["hex_x"][0],
                "bar_t": p["bar_y"][0] - (p["hex_y"][1] + 1), "bar_b": p["bar_y"][1] + 1 - (p["hex_y"][1] + 1),
                "hex_w": p["hex_x"][1] + 1 - p["hex_x"][0], "hex_h": p["hex_y"][1] + 1 - p["hex_y"][0]}
        for k in model:
            # the panels are not all at exactly scale 1 (the hexagon is 405..407 wide for 400 + stroke): 1 % on the
            # two absolute sizes, 3.5 units on every position relative to the hexagon
            assert abs(model[k] - shot[k]) <= (5.0 if k in ("hex_w", "hex_h") else 3.5), (name, k, model[k], shot[k])
        assert p["text_grey_max"] == 128                    # .5 grey
        assert p["bar_grey_mode"] in (84, 168)              # .33 / .66 grey
