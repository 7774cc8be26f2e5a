"""The image observation's model (oracle/render_np.py over oracle/cairo_model.c) against frames drawn by the REFERENCE's
real renderer -- SRC/draw.cpp + SRC/wireframe.cpp against cairo 1.16, through oracle/_ref/libsfrefdraw.so; fixtures made by
tests/golden/frames/make_frames_golden.py and make_score_golden.py.  Bar: BIT-EXACT on EVERY pixel, the score text's rows
included (round 6: the text is a glyph atlas taken from the image's real cairo + FreeType and held to 2 520 frames of the
reference's renderer: score_glyphs.npz, scores.npz).  No GPU; the HIP frames are held to the same fixtures in
tests/test_gpu_image.py."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

FRAMES = os.path.join(GOLDEN, "frames")


def _load(name):
    z = np.load(os.path.join(FRAMES, name))
    return z, json.loads(str(z["meta"]))


@pytest.mark.parametrize("name", ["scenarios.npz", "poses.npz"])
def test_model_equals_reference_frames(name):
    from oracle import render_np as R
    z, meta = _load(name)
    assert meta["cairo"] == "1.16.0"
    hb, hs = z["hex_points"][:12], z["hex_points"][12:]
    frames, snaps = z["frames"], z["snaps"]
    assert frames.shape[1:] == (92, 90) and len(frames) == len(snaps) > 600
    bad = []
    for i in range(len(frames)):
        got = R.render_raw(snaps[i], hb, hs)  # all 92 rows: the text comes from the glyph atlas
        if not np.array_equal(got, frames[i]):
            bad.append((str(z["labels"][i]), int(np.abs(got.astype(int) - frames[i].astype(int)).max())))
    assert not bad, bad[:10]


def test_score_text_equals_the_references_own_renderer():
    """scores.npz: 2 200 scores (every first / last character pair, negative ones) on a quiet state -- rows 0..8 -- and 320 frames
    with explosion rings, ships, missiles and shells UNDER the text -- whole frames: atlas + placement + pixman's OVER."""
    from oracle import render_np as R
    z, meta = _load("scores.npz")
    hb, hs = z["hex_points"][:12], z["hex_points"][12:]
    A = R.load_glyphs(0)
    assert tuple(A["layout"]) == (4, 4, 4, 1) and (A["x0"] == 31).all() and meta["grey"] == 128
    pts, rows = z["points"], z["rows"]
    assert len(pts) >= 2000 and rows.shape[1:] == (9, 90)
    assert {("%07d" % p)[0] + ("%07d" % p)[-1] for p in pts} >= {a + b for a in "0123456789-" for b in "0123456789"}
    quiet = R.render_raw(z["base"], hb, hs, text=False)
    for p, r in zip(pts, rows):
        assert np.array_equal(R.score_text_atlas(quiet, int(p), A)[:9], r), int(p)
    under = 0
    for s, f in zip(z["snaps"], z["frames"]):
        bare = R.render_raw(s, hb, hs, text=False)
        under += int(((bare[1:5, 31:59] > 0) & (f[1:5, 31:59] != bare[1:5, 31:59])).sum())
        assert np.array_equal(R.render_raw(s, hb, hs), f)
    assert under > 2000  # glyph pixels composited over something that is not black


def test_the_segment_fallback_is_not_the_references_text():
    """The seven-segment model is a named fallback (a geometry / font without an atlas): it must never stand in for the atlas
    silently -- on the default geometry the two differ by dozens of grey levels."""
    from oracle import render_np as R
    z, _ = _load("scores.npz")
    hb, hs = z["hex_points"][:12], z["hex_points"][12:]
    a = R.render_raw(z["base"], hb, hs).astype(int)
    b = R.render_raw(z["base"], hb, hs, text="segments").astype(int)
    assert np.abs(a - b)[:9].max() > 50 and np.array_equal(a[9:], b[9:])


def test_fixtures_cover_what_the_renderer_can_draw():
    z, _ = _load("poses.npz")
    lab = [str(x) for x in z["labels"]]
    S = z["snaps"]
    assert sum(x.startswith("ship_heading_") for x in lab) == 360
    assert sorted(int(S["fort_angle"][i]) for i, x in enumerate(lab) if x.startswith("fort_heading_")) == list(range(0, 360, 10))
    assert (S["ship_alive"] == 0).sum() >= 24 and (S["fort_alive"] == 0).sum() >= 6
    assert set(int(v) for v in S["vlner"]) >= set(range(14))
    assert (S["missile_alive"].sum(1) == 20).any() and (S["shell_alive"].sum(1) == 20).any()
    zs, _ = _load("scenarios.npz")
    names = {str(x).split("@")[0] for x in zs["labels"]}
    assert len(names) == 15, names
    # the frames are not trivially alike: thousands of distinct pixels light up across the set
    assert (zs["frames"].max(0) > 0).sum() > 3000


def test_model_equals_reference_frames_in_other_geometries():
    """SSF_Env(scale, viewport, ls) (ENV:50-60): surface int(vw * scale) x int(vh * scale), scale_x = w / vw and scale_y =
    h / vh separately (SRC/draw.cpp:70-71) -- the third geometry's 450 * .25 is not whole."""
    from oracle import render_np as R
    z, meta = _load("geometries.npz")
    hb, hs = z["hex_points"][:12], z["hex_points"][12:]
    snaps = z["snaps"]
    for gi, (sc, vx, vy, vw, vh, ls) in enumerate(z["geometries"]):
        prev = R.set_geometry(sc, (vx, vy, vw, vh), ls)
        try:
            frames = z["frames_%d" % gi]
            assert frames.shape[1:] == (R.H, R.W) == (int(vh * sc), int(vw * sc))
            A = R.load_glyphs(gi + 1)  # this geometry's glyphs, from the same cairo (make_score_golden.py)
            assert np.array_equal(A["geometry"], [sc, vx, vy, vw, vh, ls])
            for i in range(len(snaps)):
                assert np.array_equal(R.render_raw(snaps[i], hb, hs, glyphs=A), frames[i]), (gi, i)  # every row
        finally:
            R.set_geometry(*prev)
    assert (R.W, R.H, R.SX, R.SY) == (90, 92, .2, .2)


def test_live_reference_renderer_if_built(oracle_mod):
    """Where oracle/_ref/libsfrefdraw.so exists (the build container): a fresh hunter run of the reference, every 5th
    frame drawn by its renderer and by the model."""
    O = oracle_mod
    if not O.have_refdraw():
        pytest.skip("oracle/_ref/libsfrefdraw.so not built here (needs /root/reference + cairo)")
    from oracle import render_np as R
    import sfscript
    for gt in ("youturn", "autoturn"):
        g = O.RefDrawGame(gt, seed=7, spawn_skip=3)
        hx = g.hex_points()
        rng = np.random.default_rng(3)
        keys = [0, 1, 2, 4, 8] if gt == "youturn" else [0, 1, 2]
        for t in range(1500):
            g.apply_keys(int(rng.choice(keys)), g.youturn)
            g.step_one_tick(34)
            if t % 5 == 0:
                s = g.snapshot()
                assert np.array_equal(R.render_raw(s, hx[:12], hx[12:]), g.draw()), (gt, t)


def test_model_equals_reference_frames_in_a_close_up():
    """zoom.npz (make_zoom_golden.py): scale 0.75 on a 250 x 260 viewport around the fortress -- every pixel."""
    from oracle import render_np as R
    z = np.load(os.path.join(GOLDEN, "frames", "zoom.npz"))
    g = z["geometry"]
    hx = z["hex_points"]
    R.set_geometry(g[0], tuple(g[1:5]), g[5])
    try:
        for i, (s, f) in enumerate(zip(z["snaps"], z["frames"])):
            assert np.array_equal(R.render_raw(s, hx[:12], hx[12:], text=False), f), i
    finally:
        R.set_geometry(.2, (130, 80, 450, 460), 3)
