"""CPU-side checks of the image observation: the library's host tables and its formulation of cairo's rasterisation
(sf_tor.h, run on the host through sf_image_object_alpha / sf_image_explosion_host) against the model
(oracle/render_np.py over oracle/cairo_model.c, itself pinned to the reference's frames and to the real cairo), and
properties of the model itself.  No GPU."""
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
    assert abs(bg.astype(np.float64).sum() / 255.0 - per * 0.2 * 0.6) < 0.03 * per * 0.2 * 0.6  # (15 sub-rows per row, not exact area)
    assert L.sf_image_background(None) < 0


def test_resize_tables_follow_opencv_area():
    from oracle import render_np as R
    L = _lib()
    for ss, ds in ((90, 84), (92, 84), (125, 84), (137, 84), (168, 84), (200, 84), (251, 84), (84, 84)):
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
        assert k == len(tab) and c.max() <= (3 if ss < 2 * ds else 4)
        assert np.allclose(a.sum(1), 1.0, atol=1e-6)
        assert f[0] == 0 and f[-1] + c[-1] == ss  # covers the source exactly
    i32 = np.zeros(4, np.int32)
    assert L.sf_resize_area_tab(252, 84, i32.ctypes.data_as(C.c_void_p), i32.ctypes.data_as(C.c_void_p),
                                i32.ctypes.data_as(C.c_void_p)) < 0  # a threefold shrink or more: five taps, not this table's layout


def test_dirty_boxes_of_the_resampling_are_exact():
    """What the render kernel's sparse resampling rests on (sf_render.hip: out_box, kReachX / kReachY), from the tap
    tables themselves: source cell s is read with a weight that is not zero by the destinations floor(N s / D) ...
    ceil(N (s + 1) / D) - 1 and by no other (N / D = 14/15 for the columns, 21/23 for the rows) -- both ends attained --
    and a destination's taps are adjacent cells, two in a row and at most three in a column: the destinations that read a
    box read nothing further than one column and two rows outside it.  (sf_create asserts the same on the device side.)"""
    L = _lib()
    for ss, ds, N, D, reach in ((90, 84, 14, 15, 1), (92, 84, 21, 23, 2)):
        f, c, a = np.zeros(ds, np.int32), np.zeros(ds, np.int32), np.zeros((ds, 4), np.float32)
        assert L.sf_resize_area_tab(ss, ds, f.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                                    a.ctypes.data_as(C.c_void_p)) == 0
        readers = [[] for _ in range(ss)]
        for d in range(ds):
            assert c[d] <= reach + 1
            for j in range(c[d]):
                if a[d, j] != 0:
                    readers[f[d] + j].append(d)
        for s in range(ss):
            lo, hi = (s * N) // D, ((s + 1) * N + D - 1) // D
            assert readers[s] and min(readers[s]) == lo and max(readers[s]) == hi - 1, (ss, s, readers[s], lo, hi)
        # the kernel's box arithmetic: destinations of the source range [s0, s1)
        for s0 in range(ss):
            for s1 in range(s0 + 1, min(s0 + 30, ss) + 1):
                want = sorted({d for s in range(s0, s1) for d in readers[s]})
                assert want == list(range((s0 * N) // D, min((s1 * N + D - 1) // D, ds))), (ss, s0, s1)
                # ... and what those destinations read lies within `reach` cells of the range
                for d in want:
                    taps = [f[d] + j for j in range(c[d]) if a[d, j] != 0]
                    assert min(taps) >= s0 - reach and max(taps) <= s1 - 1 + reach, (ss, s0, s1, d, taps)


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


def test_host_tables_of_another_geometry_match_model():
    """sf_set_image_geometry's host tables -- the hexagons under scale(s) translate(-vx, -vy) with another line width, and
    INTER_AREA from a surface that is not 90 x 92 -- against the numpy model parametrised the same way."""
    from oracle import render_np as R
    L = _lib()
    hb, hs = _hex()
    rng = np.random.default_rng(4)
    for scale, vp, ls in ((.25, (100, 60, 500, 520), 2), (.3, (130, 80, 450, 460), 4.5), (.25, (130, 80, 450, 460), 3), (.4, (130, 80, 450, 460), 3)):
        prev = R.set_geometry(scale, vp, ls)
        try:
            w, h = int(vp[2] * scale), int(vp[3] * scale)
            assert (R.W, R.H) == (w, h)
            bg = np.zeros((h, w), np.uint8)
            assert L.sf_image_background_geom(w, h, vp[0], vp[1], vp[2], vp[3], ls, bg.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(bg, R.background(hb, hs)) and bg.max() > 100
            for img in (bg, rng.integers(0, 256, (h, w)).astype(np.uint8)):
                out = np.zeros((84, 84), np.uint8)
                assert L.sf_resize_area_u8(img.ctypes.data_as(C.c_void_p), w, h, out.ctypes.data_as(C.c_void_p), 84, 84) == 0
                assert np.array_equal(out, R.resize_area(img))
        finally:
            R.set_geometry(*prev)
    assert (R.W, R.H, R.SCALE, R.LINE_W) == (90, 92, .2, 3.0)


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
    assert d[1:5, 31:59].sum() > 128 * 20           # seven zeros
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
            want = R.score_text_atlas(want, 0, R.load_glyphs(0))  # the reference's glyphs (score_glyphs.npz)
        if v & 2:
            want = R.bar_frame(want, 0, False)
        assert np.array_equal(out, want), v
    assert L.sf_image_static(4, bg.ctypes.data_as(C.c_void_p)) < 0


def test_builtin_glyph_atlas_is_the_fixture():
    """The atlas compiled into the library (sf_glyphs.h) = tests/golden/frames/score_glyphs.npz geometry 0, which
    make_score_golden.py took from the image's cairo + FreeType and held to the reference renderer's frames."""
    from oracle import render_np as R
    from spacefortress_amd._lib import ScoreGlyphs
    L = _lib()
    g, a = ScoreGlyphs(), np.zeros((11, 4, 4), np.uint8)
    assert L.sf_default_score_glyphs(C.byref(g), a.ctypes.data_as(C.c_void_p), a.size) == 0
    A = R.load_glyphs(0)
    assert (g.gw, g.gh, g.advance, g.y0) == tuple(int(v) for v in A["layout"])
    assert np.array_equal(np.array([[g.x0[i][j] for j in range(10)] for i in range(11)]), A["x0"])
    assert np.array_equal(a, A["alpha"])
    assert L.sf_default_score_glyphs(C.byref(g), a.ctypes.data_as(C.c_void_p), 10) < 0


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


def test_explosion_geometry_matches_the_references_own_screenshot():
    """The right panel of rl/imgs/screens.png is a fortress explosion rendered by the reference: every yellow /
    red pixel of it must lie on one of the model's arcs (or on the radius-7 circle) of the right colour class,
    and every one of the model's 84 arcs must have pixels of the picture on it (radii within 2.5 user units,
    angles within 2 units of arc length)."""
    from oracle import render_np as R
    z = np.load(os.path.join(GOLDEN, "telemetry", "explosion_pixels.npz"))
    arcs = R.explosion_arcs()
    hit = np.zeros(len(arcs), int)
    stray = 0
    for colour, grey in (("yellow", 191), ("red", 128)):
        for r, th in zip(z[colour + "_r"], z[colour + "_theta"]):
            if colour == "yellow" and abs(r - 7) <= 2.5:
                continue  # the circle
            slack = np.degrees(2.0 / r)
            ok = False
            for k, (radius, a0, a1, g) in enumerate(arcs):
                if g != grey or abs(r - radius) > 2.5:
                    continue
                d = (th - a0) % 360
                if d <= (a1 - a0) + slack or d >= 360 - slack:
                    hit[k] += 1
                    ok = True
            stray += not ok
    n = len(z["yellow_r"]) + len(z["red_r"])
    assert stray <= 0.01 * n, (stray, n)
    assert (hit > 0).all(), np.flatnonzero(hit == 0)
    assert (np.abs(z["yellow_r"] - 7) <= 2.5).sum() > 20  # the circle is there too


def test_fortress_wireframe_matches_the_references_own_screenshot():
    """The live fortress of the left and middle panels of rl/imgs/screens.png: for one heading that is a multiple
    of the 10-degree sector, (nearly) every yellow pixel lies within 2 user units of a segment of the model's
    fortress wireframe, and every segment is covered along its length."""
    from oracle import render_np as R
    z = np.load(os.path.join(GOLDEN, "telemetry", "fortress_pixels.npz"))

    def dist_to_segment(p, a, b):
        ab, ap = b - a, p - a
        t = np.clip((ap @ ab) / (ab @ ab), 0, 1)
        return np.hypot(*(ap - np.outer(t, ab)).T), t

    for name in ("left", "middle"):
        pts = z[name].astype(np.float64)
        assert len(pts) > 60
        best = None
        for ang in range(0, 360, 10):
            c, s = np.cos(np.radians(ang)), np.sin(np.radians(ang))
            d = np.full(len(pts), 1e9)
            cover = []
            for ax, ay, bx, by in R.FORT_LINES:
                a = np.array([c * ax - s * ay, s * ax + c * ay])
                b = np.array([c * bx - s * by, s * bx + c * by])
                dk, tk = dist_to_segment(pts, a, b)
                near = dk <= 2.0
                cover.append((tk[near].min(), tk[near].max()) if near.any() else (1, 0))
                d = np.minimum(d, dk)
            frac = (d <= 2.0).mean()
            if best is None or frac > best[0]:
                best = (frac, ang, cover)
        frac, ang, cover = best
        assert frac >= 0.97, (name, frac, ang)
        for lo, hi in cover:  # each of the four segments is drawn end to end
            assert lo <= 0.15 and hi >= 0.85, (name, ang, cover)


def test_ship_wireframe_matches_the_references_own_screenshot():
    """The ship of each panel of rl/imgs/screens.png (position and heading unknown): for some integer heading and
    a position near the pixels' centroid, (nearly) every yellow pixel lies within 2 user units of a segment of
    the model's ship wireframe and every segment is covered end to end."""
    from oracle import render_np as R
    z = np.load(os.path.join(GOLDEN, "telemetry", "ship_pixels.npz"))
    lines = np.array(R.SHIP_LINES, np.float64)
    lens = np.hypot(lines[:, 2] - lines[:, 0], lines[:, 3] - lines[:, 1])
    mid = np.stack([(lines[:, 0] + lines[:, 2]) / 2, (lines[:, 1] + lines[:, 3]) / 2], 1)
    c_local = (mid * lens[:, None]).sum(0) / lens.sum()  # centroid of the strokes in wireframe coordinates

    def score(pts, ang, pos):
        c, s = np.cos(np.radians(ang)), np.sin(np.radians(ang))
        d = np.full(len(pts), 1e9)
        cover = []
        for ax, ay, bx, by in lines:
            a = pos + np.array([c * ax - s * ay, s * ax + c * ay])
            b = pos + np.array([c * bx - s * by, s * bx + c * by])
            ab, ap = b - a, pts - a
            t = np.clip((ap @ ab) / (ab @ ab), 0, 1)
            dk = np.hypot(*(ap - np.outer(t, ab)).T)
            near = dk <= 2.0
            cover.append((t[near].min(), t[near].max()) if near.any() else (1, 0))
            d = np.minimum(d, dk)
        return (d <= 2.0).mean(), cover

    for name in ("left", "middle", "right"):
        pts = z[name].astype(np.float64)
        assert len(pts) > 60
        cen = pts.mean(0)
        best = (0, None, None)
        for ang in range(360):
            c, s = np.cos(np.radians(ang)), np.sin(np.radians(ang))
            pos0 = cen - np.array([c * c_local[0] - s * c_local[1], s * c_local[0] + c * c_local[1]])
            f, cover = score(pts, ang, pos0)
            if f > best[0]:
                best = (f, ang, pos0)
        f, ang, pos0 = best
        fine = max((score(pts, a2, pos0 + np.array([dx, dy])) + (a2,) for a2 in (ang - 1, ang, ang + 1)
                    for dx in (-2, -1, 0, 1, 2) for dy in (-2, -1, 0, 1, 2)), key=lambda r: r[0])
        assert fine[0] >= 0.97, (name, fine[0], fine[2])
        for lo, hi in fine[1]:
            assert lo <= 0.15 and hi >= 0.85, (name, fine)


def test_fast_divmod_of_the_pixel_loops_is_exact():
    """sf_render.hip: fast_divmod -- i / w as (int)((float)i + 0.5f) * rw) with rw = v_rcp_f32(w), which is allowed to be an
    ulp off.  Exhaustive over every index of a box of the 90x92 surface and every box width, for 1 / w rounded to
    nearest and for its two neighbours (float32 arithmetic restated with numpy)."""
    i = np.arange(0, 90 * 92, dtype=np.int64)
    fi = i.astype(np.float32) + np.float32(0.5)
    for w in range(1, 93):
        r0 = np.float32(1.0) / np.float32(w)
        for rw in (r0, np.nextafter(r0, np.float32(0)), np.nextafter(r0, np.float32(2))):
            q = (fi * np.float32(rw)).astype(np.int64)  # float32 product, truncation
            assert np.array_equal(q, i // w), (w, rw)


def test_lerp_of_white_is_the_alpha():
    """cairo's 8-bit lerp (cairo-image-compositor.c: mul8x2_8, + 0x7f): white through alpha a onto black is a."""
    mul = lambda a, b: (((a * b + 0x7f) + ((a * b + 0x7f) >> 8)) >> 8) & 0xff
    for a in range(256):
        assert mul(255, a) == a


def _object_alpha(L, kind, x, y, ang, geom=(90, 92, 130., 80., 450., 460., 3.0)):
    w, h = geom[0], geom[1]
    got = np.zeros((h, w), np.uint8)
    assert L.sf_image_object_alpha(kind, x, y, ang, w, h, *geom[2:], got.ctypes.data_as(C.c_void_p)) == 0
    return got


def test_kernel_formulation_of_a_wireframe_equals_the_model():
    """sf_tor.h -- an object as convex quads united by inclusion-exclusion, a pixel row sampled in 15 sub-rows or taken whole,
    cairo's integer quotients carried as doubles -- against the edge-list restatement (oracle/cairo_model.c), bit for bit:
    every kind of wireframe at random poses, whole-pixel and quarter-pixel positions, the diagonal and axis headings where
    edges tie, and across the surface's four borders (cairo clips the polygon there: extra vertices)."""
    from oracle import render_np as R
    L = _lib()
    lines = [R.SHIP_LINES, R.FORT_LINES, R.MISSILE_LINES, R.SHELL_LINES]
    rng = np.random.default_rng(5)
    for it in range(6000):
        kind, mode = it % 4, it % 5
        if mode == 0:
            side, t = rng.integers(0, 4), rng.uniform(0, 1)
            x, y = [(130 + rng.uniform(-7, 7), 80 + 460 * t), (580 + rng.uniform(-7, 7), 80 + 460 * t),
                    (130 + 450 * t, 80 + rng.uniform(-7, 7)), (130 + 450 * t, 540 + rng.uniform(-7, 7))][side]
        elif mode == 1:
            x, y = float(rng.integers(125, 585)), float(rng.integers(75, 545))
        elif mode == 2:
            x, y = rng.integers(125 * 4, 585 * 4) / 4.0, rng.integers(75 * 4, 545 * 4) / 4.0
        else:
            x, y = rng.uniform(120, 590), rng.uniform(70, 550)
        ang = int(rng.integers(0, 360)) if it % 3 else int(rng.choice([0, 45, 90, 135, 180, 225, 270, 315, 30, 60]))
        if kind == 1 and ang == 0:
            ang = 10  # (the rectilinear stroker's pose: sf_image_fort_alpha below)
        want = R.run_script(R.s_begin() + R.s_wireframe(lines[kind], (x, y), ang))
        assert np.array_equal(_object_alpha(L, kind, x, y, ang), want), (kind, x, y, ang)


def test_kernel_formulation_in_other_geometries():
    from oracle import render_np as R
    L = _lib()
    lines = [R.SHIP_LINES, R.FORT_LINES, R.MISSILE_LINES, R.SHELL_LINES]
    rng = np.random.default_rng(6)
    for scale, vp, ls in ((.25, (100, 60, 500, 520), 2), (.3, (130, 80, 450, 460), 4.5), (.25, (130, 80, 450, 460), 3)):
        prev = R.set_geometry(scale, vp, ls)
        try:
            for it in range(400):
                kind = it % 4
                x, y, ang = rng.uniform(vp[0] - 5, vp[0] + vp[2] + 5), rng.uniform(vp[1] - 5, vp[1] + vp[3] + 5), int(rng.integers(1, 360))
                want = R.run_script(R.s_begin() + R.s_wireframe(lines[kind], (x, y), ang))
                got = _object_alpha(L, kind, x, y, ang, (R.W, R.H, float(vp[0]), float(vp[1]), float(vp[2]), float(vp[3]), float(ls)))
                assert np.array_equal(got, want), (scale, kind, x, y, ang)
        finally:
            R.set_geometry(*prev)


def test_kernel_formulation_of_the_explosion_equals_the_model():
    """drawExplosion: 84 arcs as single quads between the faces cairo's stroker puts at their ends, the circle as sixteen
    pieces of two flattened Bezier halves."""
    from oracle import render_np as R
    L = _lib()
    rng = np.random.default_rng(7)
    for it in range(400):
        x, y = (355.0, 315.0) if it == 0 else (rng.uniform(125, 585), rng.uniform(75, 545))
        if it % 5 == 1:
            x, y = float(int(x)), float(int(y))
        want = R.run_script(R.s_begin() + R.s_explosion((x, y)))
        got = np.zeros((92, 90), np.uint8)
        assert L.sf_image_explosion_host(x, y, 90, 92, 130., 80., 450., 460., 3.0, got.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(got, want), (x, y)


def test_explosions_in_other_geometries_flatten_like_cairo():
    """At a larger scale cairo cuts a 10-degree arc into two pieces (its control points lie further than the tolerance 0.1
    pixel from the chord) and the circle's halves into sixteen: the adaptive form (sft::flatten_faces), where the faces
    BETWEEN the pieces bound them without being edges of the polygon."""
    from oracle import render_np as R
    L = _lib()
    rng = np.random.default_rng(8)
    pieces = set()
    for scale, vp, ls in ((.25, (100, 60, 500, 520), 2), (.3, (130, 80, 450, 460), 4.5), (.4, (130, 80, 450, 460), 3), (.55, (130, 80, 450, 460), 3)):
        prev = R.set_geometry(scale, vp, ls)
        g = (R.W, R.H, float(vp[0]), float(vp[1]), float(vp[2]), float(vp[3]), float(ls))
        try:
            for it in range(60):
                x, y = rng.uniform(vp[0] - 5, vp[0] + vp[2] + 5), rng.uniform(vp[1] - 5, vp[1] + vp[3] + 5)
                want = R.run_script(R.s_begin() + R.s_explosion((x, y)))
                got = np.zeros((R.H, R.W), np.uint8)
                assert L.sf_image_explosion_host(x, y, *g, got.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(got, want), (scale, x, y)
            for it in range(200):
                x, y, r = rng.uniform(vp[0], vp[0] + vp[2]), rng.uniform(vp[1], vp[1] + vp[3]), rng.uniform(10, 70)
                a1 = rng.uniform(0, 6.2)
                a2 = a1 + rng.uniform(.05, 1.5)
                want = R.run_script(R.s_begin() + [R.LINE_WIDTH, float(ls), R.GREY, 1.0, R.ARC, x, y, r, a1, a2, R.STROKE])
                got = np.zeros((R.H, R.W), np.uint8)
                n = L.sf_image_arc_alpha(x, y, r, a1, a2, *g, got.ctypes.data_as(C.c_void_p))
                assert n >= 1 and np.array_equal(got, want), (scale, x, y, r, a1, a2)
                pieces.add(n)
        finally:
            R.set_geometry(*prev)
    assert {1, 2, 4} <= pieces, pieces


def test_fortress_alpha_maps_equal_the_model():
    """The 36 pictures of the live fortress the frame kernel lerps in or starts from: heading 0 through cairo's rectilinear
    stroker and box converter, the others through the scan converter."""
    from oracle import render_np as R
    L = _lib()
    for sector in range(36):
        a = np.zeros((16, 16), np.uint8)
        assert L.sf_image_fort_alpha(sector, a.ctypes.data_as(C.c_void_p)) == 0
        want = R.run_script(R.s_begin() + R.s_wireframe(R.FORT_LINES, R.FORT, 10 * sector))
        assert np.array_equal(a, want[39:55, 37:53]) and want.sum() == want[39:55, 37:53].sum(), sector
    assert L.sf_image_fort_alpha(36, a.ctypes.data_as(C.c_void_p)) < 0
