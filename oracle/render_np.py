"""TEST INFRASTRUCTURE ONLY -- the image observation, restated: what the reference draws and what cairo makes of it.

What is drawn, where, in which order and in which grey comes from the reference's renderer
(SRC = /root/reference/python/spacefortress/src):

  drawGameStateScaled   SRC/draw.cpp:256-270   scale(w / vp_w, h / vp_h), translate(-vp_x, -vp_y), black paint
  drawJustGameStuff     SRC/draw.cpp:227-254   hexagons, ship|explosion, fortress|explosion,
                                               missiles, shells (only when > 21 from the fortress)
  drawWireFrame         SRC/draw.cpp:82-100    translate, rotate(int angle), lines as separate sub-paths, ONE stroke
  drawHexagon           SRC/draw.cpp:102-114   closed path, miter joins
  wireframes            SRC/wireframe.cpp:11-67
  drawExplosion         SRC/draw.cpp:116-145   84 arcs, each its own stroke, and the radius-7 circle
  drawScore             SRC/draw.cpp:161-173   "%07d", grey .5
  drawVlner             SRC/draw.cpp:207-225   two filled rectangles
  SSF_Env._draw         ENV:203-206            grey channel of the RGB24 surface, [92][90] uint8
  WrapPyTorch           rl/envs.py:28-30       cv2.resize(.., (84, 84), INTER_AREA)

HOW a stroke becomes grey levels is cairo's image backend (1.16.0 in this image).  oracle/cairo_model.c restates it
(stroker, 15-sub-row scan converter with its full-row shortcut, box converter, 8-bit lerp) and is pinned bit for bit to the
real library (tests/test_cairo_model.py) and to frames of the reference's real draw.cpp (tests/golden/frames/, made by
make_frames_golden.py through oracle/_ref/libsfrefdraw.so): this module builds the draw script of a frame and runs it through
that model.  The score text (drawScore, SRC/draw.cpp:147-173) is cairo's toy font: FreeType A8 glyph bitmaps blitted at
whole-pixel origins and composited with pixman's OVER -- data plus a placement rule, taken from the image's real cairo and
held to frames of the reference's own renderer (tests/golden/frames/make_score_golden.py -> score_glyphs.npz, scores.npz):
`score_text_atlas`.  PINNED: every pixel, text included, on a box whose fontconfig resolves "monospace bold" to the font of
this image (DejaVu Sans Mono Bold).  The seven-segment glyph model of earlier rounds (`score_text`) survives as the named
fallback for a geometry / font without an atlas (`text="segments"`); it equals no reference pixels and is never compared
with any.  cv2 is not in this image: INTER_AREA follows OpenCV's published algorithm
(imgproc/resize.cpp, computeResizeAreaTab + resizeArea_).
"""
import ctypes as _C
import os as _os
import subprocess as _sp
import math

import numpy as np

W, H, OUT = 90, 92, 84
VP_X, VP_Y, VP_W, VP_H, SCALE, LINE_W = 130.0, 80.0, 450.0, 460.0, 0.2, 3.0
SX, SY = W / VP_W, H / VP_H  # newPixelBuffer: scale_x = width / vp_width, scale_y = height / vp_height (SRC/draw.cpp:70-71)
FORT = (355.0, 315.0)
TEXT_ROWS = 9  # rows 0..8 of the default 92 x 90 surface hold the score text (compared like every other row since round 6)

HERE = _os.path.dirname(_os.path.abspath(__file__))
MODEL_SO = _os.path.join(HERE, "libsfcairomodel.so")
GLYPHS_NPZ = _os.path.join(_os.path.dirname(HERE), "tests", "golden", "frames", "score_glyphs.npz")
GLYPH_CHARS = "0123456789-"
_model = None

# the draw-script vocabulary of oracle/cairo_model.h
(END, SAVE, RESTORE, SCALE_OP, TRANSLATE, ROTATE, LINE_WIDTH, GREY, MOVE_TO, LINE_TO, CLOSE, ARC, RECT, STROKE, FILL, PAINT,
 CURVE_TO, NEW_PATH) = range(18)


def model_lib():
    global _model
    if _model is None:
        src = _os.path.join(HERE, "cairo_model.c")
        if not _os.path.exists(MODEL_SO) or (_os.path.exists(src) and _os.path.getmtime(MODEL_SO) < _os.path.getmtime(src)):
            _sp.check_call(["make", "-C", HERE, "liboracle"], stdout=_sp.DEVNULL)
        L = _C.CDLL(MODEL_SO)
        for f in ("cm_run", "cm_run_over"):
            getattr(L, f).argtypes = [_C.c_void_p, _C.c_int, _C.c_int, _C.c_int, _C.c_void_p]
        L.cm_last_polygon.argtypes = [_C.c_void_p, _C.c_int]
        _model = L
    return _model


def run_script(script, w=None, h=None, onto=None):
    """The script through oracle/cairo_model.c: a fresh black surface, or composited onto a copy of `onto`."""
    w, h = W if w is None else w, H if h is None else h
    s = np.ascontiguousarray(script, np.float64)
    out = np.zeros((h, w), np.uint8) if onto is None else np.ascontiguousarray(onto, np.uint8).copy()
    rc = (model_lib().cm_run if onto is None else model_lib().cm_run_over)(
        s.ctypes.data_as(_C.c_void_p), len(s), w, h, out.ctypes.data_as(_C.c_void_p))
    assert rc == 0, rc
    return out


def set_geometry(scale=.2, viewport=(130, 80, 450, 460), ls=3):
    """The geometry SSF_Env(scale, viewport, ls) asks for (ENV:50-60): surface int(vw * scale) x int(vh * scale)
    (pymodule.cpp:351 -> newPixelBuffer), device = (user - (vx, vy)) * (w / vw, h / vh) -- NOT `scale` itself when vw * scale
    is not whole (SRC/draw.cpp:70-71,259) --, line width ls user units.  Module-wide (the functions below read the globals);
    returns the previous (scale, viewport, ls) so that a test can put it back."""
    global W, H, VP_X, VP_Y, VP_W, VP_H, SCALE, LINE_W, SX, SY
    prev = (SCALE, (VP_X, VP_Y, VP_W, VP_H), LINE_W)
    VP_X, VP_Y, VP_W, VP_H = (float(v) for v in viewport)
    SCALE, LINE_W = float(scale), float(ls)
    W, H = int(VP_W * SCALE), int(VP_H * SCALE)
    SX, SY = W / VP_W, H / VP_H
    return prev


SHIP_LINES = [(-18, 0, 18, 0), (-18, 18, 0, 0), (0, 0, -18, -18)]
FORT_LINES = [(0, 0, 36, 0), (0, -18, 18, -18), (18, -18, 18, 18), (18, 18, 0, 18)]
MISSILE_LINES = [(0, 0, -25, 0), (0, 0, -5, 5), (0, 0, -5, -5)]
SHELL_LINES = [(-8, 0, 0, -6), (0, -6, 16, 0), (16, 0, 0, 6), (0, 6, -8, 0)]

# seven-segment glyph model of the score text (see sf_raster.h)
TXT_ADV, TXT_PAD, TXT_W, TXT_H, TXT_T = 18.0, 3.0, 16.0, 22.0, 5.0
TXT_X0 = 355.0 - 3.5 * TXT_ADV
TXT_TOP = 97.0 - 0.5 * TXT_H
SEGS = {"0": "ABCDEF", "1": "BC", "2": "ABDEG", "3": "ABCDG", "4": "BCFG", "5": "ACDFG", "6": "ACDEFG",
        "7": "ABC", "8": "ABCDEFG", "9": "ABCDFG", "-": "G"}

M_PI = 3.14159265358979323846


def deg2rad(a):
    """SRC/vector.cpp:34-36"""
    return a * M_PI / 180.0


# ---- draw scripts: the calls of SRC/draw.cpp, one for one ---------------------------------------------------------------
def s_begin():
    """drawGameStateScaled up to the paint, SRC/draw.cpp:256-263"""
    return [SCALE_OP, SX, SY, TRANSLATE, -VP_X, -VP_Y, LINE_WIDTH, LINE_W, GREY, 0.0, PAINT]


def s_hexagon(pts, grey=1.0):
    """drawHexagon, SRC/draw.cpp:102-114 (the line width is the context's: set in drawGameStateScaled)"""
    pts = np.asarray(pts, np.float64).reshape(6, 2)
    s = [GREY, grey, MOVE_TO, pts[0, 0], pts[0, 1]]
    for x, y in pts[1:]:
        s += [LINE_TO, x, y]
    return s + [CLOSE, STROKE]


def s_wireframe(lines, pos, angle, grey=1.0):
    """drawWireFrame, SRC/draw.cpp:82-100: `angle` arrives as an int (a double heading is truncated by the call)"""
    s = [SAVE, TRANSLATE, float(pos[0]), float(pos[1]), ROTATE, deg2rad(int(angle)), LINE_WIDTH, LINE_W, GREY, grey]
    for ax, ay, bx, by in lines:
        s += [MOVE_TO, ax, ay, LINE_TO, bx, by]
    return s + [STROKE, RESTORE]


def explosion_arcs():
    """(radius, start degree, end degree, grey) of the 84 arcs of drawExplosion, SRC/draw.cpp:121-137."""
    arcs, ofs = [], 0
    for radius in range(15, 70, 8):
        ofs += 3
        grey = 191 if radius < 60 else 128  # .75 (yellow in colour mode), .5 (red)
        for angle in range(0, 360, 30):
            arcs.append((radius, angle + ofs, angle + ofs + 10, grey))
    return arcs


def s_explosion(pos):
    """drawExplosion, SRC/draw.cpp:116-145: every arc is stroked on its own; the line width STAYS set for what follows"""
    x, y = float(pos[0]), float(pos[1])
    s = [LINE_WIDTH, float(np.float32(LINE_W))]  # `float ls`
    for radius, a0, a1, grey in explosion_arcs():
        s += [GREY, .75 if grey == 191 else .5, ARC, x, y, float(radius), deg2rad(a0), deg2rad(a1), STROKE]
    return s + [GREY, .75, ARC, x, y, 7.0, 0.0, M_PI * 2, STROKE]


def s_bar(vlner, kill):
    """drawVlner, SRC/draw.cpp:207-225"""
    return [LINE_WIDTH, float(np.float32(LINE_W)) - 1, GREY, .33, RECT, 355.0 - 100, 335.0 + 187, 200.0, 10.0, FILL,
            GREY, 1.0 if kill else .66, RECT, 355.0 - 100, 335.0 + 187, float(20 * (10 if vlner > 10 else vlner)), 10.0, FILL]


def s_objects(snap):
    """drawJustGameStuff after the hexagons, SRC/draw.cpp:233-253"""
    s = []
    ship = (float(snap["ship_x"]), float(snap["ship_y"]))
    s += s_wireframe(SHIP_LINES, ship, snap["ship_angle"]) if snap["ship_alive"] else s_explosion(ship)
    s += s_wireframe(FORT_LINES, FORT, snap["fort_angle"]) if snap["fort_alive"] else s_explosion(FORT)
    for i in range(len(snap["missile_alive"])):
        if snap["missile_alive"][i]:
            s += s_wireframe(MISSILE_LINES, (snap["missile_x"][i], snap["missile_y"][i]), snap["missile_angle"][i])
    for i in range(len(snap["shell_alive"])):
        d = math.sqrt((snap["shell_x"][i] - FORT[0]) ** 2 + (snap["shell_y"][i] - FORT[1]) ** 2)
        if snap["shell_alive"][i] and d > 21:
            s += s_wireframe(SHELL_LINES, (snap["shell_x"][i], snap["shell_y"][i]), snap["shell_angle"][i])
    return s


# ---- the score text: the seven-segment FALLBACK model (equals no reference pixels, see the module docstring) ------------------------------------------------
def _dev(pts):
    pts = np.asarray(pts, np.float64)
    return np.stack([(pts[:, 0] - VP_X) * SX, (pts[:, 1] - VP_Y) * SY], 1)


def _clip_unit(poly, e):
    """One Sutherland-Hodgman pass against edge e of the unit square: x >= 0, x <= 1, y >= 0, y <= 1."""
    out = []
    n = len(poly)
    for i in range(n):
        (x0, y0), (x1, y1) = poly[i], poly[(i + 1) % n]
        d0, d1 = ((x0, x1), (1.0 - x0, 1.0 - x1), (y0, y1), (1.0 - y0, 1.0 - y1))[e]
        if d0 >= 0:
            out.append((x0, y0))
        if (d0 >= 0) != (d1 >= 0):
            t = d0 / (d0 - d1)
            out.append((x0 + t * (x1 - x0), y0 + t * (y1 - y0)))
    return out


def pixel_area(poly, px, py):
    """Area of convex `poly` inside pixel [px,px+1] x [py,py+1] (Sutherland-Hodgman + shoelace)."""
    p = [(float(x) - px, float(y) - py) for x, y in poly]
    for e in range(4):
        p = _clip_unit(p, e)
        if not p:
            return 0.0
    s = 0.0
    n = len(p)
    for i in range(n):
        (x0, y0), (x1, y1) = p[i], p[(i + 1) % n]
        s += x0 * y1 - x1 * y0
    return 0.5 * abs(s)


def mul_un8(a, b):
    t = a * b + 128
    return (t + (t >> 8)) >> 8


def over(fb, poly, grey):
    """Composite convex polygon `poly` (device coordinates) in `grey` OVER the uint8 frame: exact area, pixman's rounding
    (the glyph model's arithmetic; everything cairo draws goes through run_script instead)."""
    poly = np.asarray(poly, np.float64)
    x0, y0 = np.floor(poly.min(0)).astype(int)
    x1, y1 = np.ceil(poly.max(0)).astype(int)
    for py in range(max(y0, 0), min(y1, fb.shape[0])):
        for px in range(max(x0, 0), min(x1, fb.shape[1])):
            a = min(pixel_area(poly, px, py), 1.0)
            m = int(a * 255.0 + 0.5)
            if m > 0:
                fb[py, px] = mul_un8(grey, m) + mul_un8(int(fb[py, px]), 255 - m)


def rect_poly(x0, y0, x1, y1):
    return _dev([(x0, y0), (x1, y0), (x1, y1), (x0, y1)])


def score_text(fb, points):
    text = "%07d" % int(points)
    m0, m1 = 0.5 * (TXT_H - TXT_T), 0.5 * (TXT_H + TXT_T)
    seg_rect = {"A": (0, 0, TXT_W, TXT_T), "B": (TXT_W - TXT_T, TXT_T, TXT_W, m0),
                "C": (TXT_W - TXT_T, m1, TXT_W, TXT_H - TXT_T), "D": (0, TXT_H - TXT_T, TXT_W, TXT_H),
                "E": (0, m1, TXT_T, TXT_H - TXT_T), "F": (0, TXT_T, TXT_T, m0), "G": (0, m0, TXT_W, m1)}
    for cell, ch in enumerate(text[:7]):
        gx = TXT_X0 + TXT_ADV * cell + TXT_PAD
        for seg in "ABCDEFG":  # the kernel's lane order: A, B, C, D, E, F, G
            if seg in SEGS[ch]:
                x0, y0, x1, y1 = seg_rect[seg]
                over(fb, rect_poly(gx + x0, TXT_TOP + y0, gx + x1, TXT_TOP + y1), 128)


# ---- frames -------------------------------------------------------------------------------------------------------------------
def background(hex_big, hex_small):
    """Both hexagons on black (drawJustGameStuff's first two calls)."""
    return run_script(s_begin() + s_hexagon(hex_big) + s_hexagon(hex_small))


def bar_frame(fb, vlner, kill):
    return run_script(s_begin()[:8] + s_bar(vlner, kill), fb.shape[1], fb.shape[0], onto=fb)


_glyphs = {}


def load_glyphs(k=0):
    """Glyph atlas `k` of tests/golden/frames/score_glyphs.npz (0: the default geometry) as dict(alpha, layout, x0, geometry)."""
    if k not in _glyphs:
        z = np.load(GLYPHS_NPZ)
        _glyphs[k] = dict(alpha=z["alpha_%d" % k], layout=z["layout_%d" % k], x0=z["x0_%d" % k], geometry=z["geometry_%d" % k])
    return _glyphs[k]


def glyphs_for_geometry():
    """The atlas whose geometry is the module's current one (set_geometry), or None."""
    z = np.load(GLYPHS_NPZ)
    cur = np.array([SCALE, VP_X, VP_Y, VP_W, VP_H, LINE_W])
    for key in z.files:
        if key.startswith("geometry_") and np.array_equal(z[key], cur):
            return load_glyphs(int(key[9:]))
    return None


def score_text_atlas(fb, points, A):
    """drawScore (SRC/draw.cpp:161-173) from a glyph atlas: "%07d", box i at (x0[first][last] + i * advance, y0), grey .5 IN the
    glyph's coverage OVER the frame (pixman: mul_un8(128, a) + mul_un8(d, 255 - a)).  Returns a new frame."""
    out = np.array(fb, np.uint8).copy()
    text = ("%07d" % int(points))[:7]
    gw, gh, adv, y0 = (int(v) for v in A["layout"])
    x0 = int(A["x0"][GLYPH_CHARS.index(text[0]), int(text[-1])])
    for i, ch in enumerate(text):
        a = A["alpha"][GLYPH_CHARS.index(ch)]
        for r in range(gh):
            for q in range(gw):
                m, y, x = int(a[r, q]), y0 + r, x0 + i * adv + q
                if m and 0 <= y < out.shape[0] and 0 <= x < out.shape[1]:
                    out[y, x] = mul_un8(128, m) + mul_un8(int(out[y, x]), 255 - m)
    return out


def render_raw(snap, hex_big, hex_small, vuln_time=250, bg=None, text=True, glyphs=None):
    """One [H][W] uint8 frame from an oracle snapshot record (oracle.SNAPSHOT_DTYPE): drawGameStateScaled's calls in its
    order -- hexagons, objects, score, bar.  `bg` (the hexagons' frame) is accepted for the callers that keep one; the
    hexagons are the first strokes on black either way.  text: True = the glyph atlas (`glyphs`, else the one recorded for
    the current geometry; an error if there is none), "segments" = the named seven-segment fallback, False = no text."""
    fb = run_script(s_begin() + s_hexagon(hex_big) + s_hexagon(hex_small) + s_objects(snap))
    if text == "segments":
        score_text(fb, snap["points"])
    elif text:
        A = glyphs if glyphs is not None else glyphs_for_geometry()
        assert A is not None, "no glyph atlas for this geometry: pass glyphs= or text='segments'"
        fb = score_text_atlas(fb, snap["points"], A)
    vlner = int(snap["vlner"])
    kill = vlner > 10 and int(snap["fort_vuln_timer"]) < vuln_time
    return bar_frame(fb, vlner, kill)


def area_tab(ssize, dsize):
    """computeResizeAreaTab: list of (di, si, alpha) in table order."""
    scale = 1.0 / (dsize / ssize)
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area(frame, dsize=(OUT, OUT)):
    """cv2.resize(frame, dsize, interpolation=INTER_AREA) for a non-integer shrink (resizeArea_)."""
    sh, sw = frame.shape
    xtab, ytab = area_tab(sw, dsize[0]), area_tab(sh, dsize[1])
    src = frame.astype(np.float32)
    out = np.zeros((dsize[1], dsize[0]), np.uint8)
    sums = {}
    for dy, sy, beta in ytab:
        buf = np.zeros(dsize[0], np.float32)
        for dx, sx, alpha in xtab:
            buf[dx] = np.float32(buf[dx] + src[sy, sx] * alpha)
        if dy in sums:
            sums[dy] = (sums[dy] + beta * buf).astype(np.float32)
        else:
            sums[dy] = (beta * buf).astype(np.float32)
    for dy, s in sums.items():
        out[dy] = np.clip(np.rint(s), 0, 255).astype(np.uint8)
    return out
