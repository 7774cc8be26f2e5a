"""oracle/cairo_model.c -- our restatement of cairo 1.16's image rasteriser, the part of the image observation that is
cairo's and not the reference's -- against the REAL library: the same draw scripts through both (oracle/cairo_probe.c ->
oracle/_ref/libcairoprobe.so, built here from the image's /opt/conda cairo; skipped where that is absent).  Random polygon
fills pin the scan converter (sub-row sampling, the full-row shortcut, clipping to the surface), random wireframes / closed
hexagons / arcs pin the stroker, rectangles the box converter; all bit-exact.  No GPU."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "oracle", "_ref", "libcairoprobe.so")


@pytest.fixture(scope="module")
def both():
    from oracle import render_np as R
    if os.path.exists("/opt/conda/include/cairo/cairo.h"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "cairoprobe"], stdout=subprocess.DEVNULL)
    if not os.path.exists(PROBE):
        pytest.skip("no cairo on this box: oracle/_ref/libcairoprobe.so not built")
    P = C.CDLL(PROBE)
    P.cp_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    P.cp_version.restype = C.c_char_p
    assert P.cp_version() == b"1.16.0"

    def run(script, w=90, h=92):
        s = np.asarray(script, np.float64)
        real = np.zeros((h, w), np.uint8)
        assert P.cp_run(s.ctypes.data_as(C.c_void_p), len(s), w, h, real.ctypes.data_as(C.c_void_p)) == 0
        return R.run_script(s, w, h), real

    return R, run


def test_random_polygon_fills(both):
    R, run = both
    rng = np.random.default_rng(0)
    for it in range(1500):
        k = rng.integers(3, 8)
        pts = rng.uniform(-6, 46, (k, 2)) if it % 3 else rng.uniform(4, 36, (k, 2))
        if it % 5 == 0:
            pts = np.round(pts * 4) / 4  # vertices on sub-row boundaries, whole pixels, shared ordinates
        if it % 11 == 0:
            pts[:, 1] = np.round(pts[:, 1])  # whole rows: the full-row shortcut with edges starting at a row's top
        s = [R.GREY, 0, R.PAINT, R.GREY, [1.0, .75, .5, .33][it % 4], R.MOVE_TO, *pts[0]]
        for p in pts[1:]:
            s += [R.LINE_TO, *p]
        got, real = run(s + [R.CLOSE, R.FILL], 40, 40)
        assert np.array_equal(got, real), (it, pts.tolist())


def test_random_wireframes(both):
    R, run = both
    rng = np.random.default_rng(1)
    for it in range(1600):
        lines = (R.SHIP_LINES, R.FORT_LINES, R.MISSILE_LINES, R.SHELL_LINES)[it % 4]
        x, y, ang = rng.uniform(120, 590), rng.uniform(70, 550), int(rng.integers(0, 360))
        if it % 4 == 1:
            ang = ang // 10 * 10
        if it % 7 == 0:
            x, y = float(int(x)), float(int(y))
        if it % 13 == 0:
            ang = int(rng.choice([0, 90, 180, 270]))  # the rectilinear special cases
        got, real = run(R.s_begin() + R.s_wireframe(lines, (x, y), ang))
        assert np.array_equal(got, real), (it, x, y, ang)
    for lw, sc in ((1.0, .2), (2.0, .25), (4.5, .3), (3.0, .5), (7.0, .2)):
        for it in range(60):
            s = [R.SCALE_OP, sc, sc * 1.01, R.TRANSLATE, -100, -50, R.GREY, 0, R.PAINT, R.SAVE, R.TRANSLATE,
                 rng.uniform(120, 400), rng.uniform(80, 350), R.ROTATE, R.deg2rad(int(rng.integers(0, 360))), R.LINE_WIDTH, lw,
                 R.GREY, 1.0]
            for ax, ay, bx, by in R.SHIP_LINES:
                s += [R.MOVE_TO, ax, ay, R.LINE_TO, bx, by]
            got, real = run(s + [R.STROKE, R.RESTORE], 120, 120)
            assert np.array_equal(got, real), (lw, sc, it)


def test_closed_hexagons_and_explosions_and_bars(both):
    R, run = both
    rng = np.random.default_rng(2)
    for it in range(200):
        r, cx, cy = (200, 355, 315) if it == 0 else (40, 355, 315) if it == 1 else (rng.uniform(20, 220), rng.uniform(300, 400), rng.uniform(280, 350))
        x1, x2, x3, x4 = math.floor(cx - r), math.floor(cx - r * .5), math.floor(cx + r * .5), math.floor(cx + r)
        y1, y3 = math.floor(cy - r * 0.8660254037844386), math.floor(cy + r * 0.8660254037844386)
        pts = [x1, cy, x2, y1, x3, y1, x4, cy, x3, y3, x2, y3]
        got, real = run(R.s_begin() + R.s_hexagon(pts))
        assert np.array_equal(got, real), (it, r, cx, cy)
    for it in range(120):
        pos = (355.0, 315.0) if it == 0 else (rng.uniform(120, 590), rng.uniform(70, 550))
        got, real = run(R.s_begin() + R.s_explosion(pos))
        assert np.array_equal(got, real), (it, pos)
    for v in range(13):
        for kill in (False, True):
            got, real = run(R.s_begin() + R.s_bar(v, kill))
            assert np.array_equal(got, real), (v, kill)
    assert got[89, 30] == 255 and real[88, 30] != real[89, 30]  # the bar's partly covered top row


def test_curves_open_polylines_and_odd_cases(both):
    """Arcs of many radii, spans and line widths (cairo-arc.c's segment count, cairo-spline.c's flattening, the tangent faces
    of cairo-path-stroke-polygon.c's spline_to), and fills of curved paths.  Domain: radius >= 7 user units, the smallest the
    reference draws (SRC/draw.cpp:144) -- below about 1.5 line widths consecutive pieces of the flattened curve turn by more
    than the stroker's cusp tolerance and cairo inserts a round pen fan, which the model does not restate (off by one level
    there, measured)."""
    R, run = both
    rng = np.random.default_rng(3)
    for it in range(300):
        r = rng.uniform(7, 120)
        a0 = rng.uniform(0, 6.3)
        a1 = a0 + rng.uniform(.05, 6.2)
        s = R.s_begin() + [R.LINE_WIDTH, 3.0 if r < 20 else rng.uniform(2, 6), R.GREY, .75, R.ARC, rng.uniform(250, 450), rng.uniform(200, 400), r, a0, a1]
        got, real = run(s + ([R.STROKE] if it % 3 else [R.CLOSE, R.FILL]))
        assert np.array_equal(got, real), (it, r, a0, a1)
