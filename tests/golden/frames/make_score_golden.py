#!/usr/bin/env python3
"""The score text as the reference draws it on this image: a glyph atlas and frames to hold it to.

Run in the build container only (needs /root/reference, the image's cairo 1.16 + FreeType + fontconfig):

    python tests/golden/frames/make_score_golden.py     ->  score_glyphs.npz, scores.npz

drawScore / centeredText (SRC/draw.cpp:147-173) draw "%07d" of the points through cairo's toy font API: "monospace" bold at 30
user units.  On the image backend that is: FreeType's A8 bitmap of every glyph (the font fontconfig resolves -- DejaVu Sans Mono
Bold here), its origin = the string's origin + i * the hinted advance, transformed to the device and rounded to whole pixels,
composited with pixman's OVER (solid grey 128 IN the glyph's coverage).  So the text is DATA plus a placement rule:

  score_glyphs.npz   per geometry `k` (0 = the default SSF_Env geometry, 1.. = those of geometries.npz):
      alpha_k  u8[11][gh][gw]   coverage of '0'..'9', '-' in one box per glyph (taken from the REAL cairo through
                                oracle/cairo_probe.c: cp_text_at -- an A8 surface, so a byte is the coverage itself)
      layout_k i32[4]           gw, gh, advance (pixels between boxes), y0 (top row of the boxes)
      x0_k     i16[11][10]      left column of the first box by (first character, last character): centeredText centres on
                                the string's INK width, which depends on its first and last glyph
      geometry_k f64[6]         scale, viewport x y w h, line width
  scores.npz         frames of the reference's OWN renderer (oracle/_ref/libsfrefdraw.so = SRC/draw.cpp compiled where it lies),
                     default geometry: `rows` u8[n][9][90] = rows 0..8 for `points` i32[n] on a quiet state (>= 2000 random
                     scores, every first / last digit pair), and `frames` u8[m][92][90] + `snaps` for states with objects UNDER
                     the text (explosion rings, missiles, shells across rows 0..8): the compositing operator.

The script asserts, before it writes anything, that atlas + rule reproduce (a) cairo's own mask of every probed string in every
geometry and (b) every one of the reference's frames it stores, all 92 rows, text included; tests/test_frames_golden.py and
tests/test_gpu_image.py repeat (b) from the files with the oracle's model and with the HIP kernels.
Nothing here is reference source: glyph coverage, integers and pixels.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import oracle as O  # noqa: E402
from oracle import render_np as R  # noqa: E402

CHARS = "0123456789-"
CX, CY = 355, 97  # centeredText(ctx, text, start, score_y - 193), SRC/draw.cpp:172
DEFAULT = (.2, (130, 80, 450, 460), 3.0)


def probe():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "cairoprobe", "refdraw"], stdout=subprocess.DEVNULL)
    P = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcairoprobe.so"))
    P.cp_score_mask.argtypes = [C.c_char_p, C.c_int, C.c_int] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    P.cp_text_at.argtypes = [C.c_char_p, C.c_int, C.c_int] + [C.c_double] * 6 + [C.c_void_p]
    return P


def surface(geom):
    sc, vp, _ = geom
    w, h = int(vp[2] * sc), int(vp[3] * sc)
    return w, h, w / vp[2], h / vp[3], float(vp[0]), float(vp[1])


def score_mask(P, geom, text):
    w, h, sx, sy, vx, vy = surface(geom)
    out, ext = np.zeros((h, w), np.uint8), np.zeros(6)
    assert P.cp_score_mask(text.encode(), w, h, sx, sy, vx, vy, CX, CY, out.ctypes.data_as(C.c_void_p), ext.ctypes.data_as(C.c_void_p)) == 0
    return out, ext


def atlas_of(P, geom):
    """(alpha[11][gh][gw], (gw, gh, adv, y0), x0[11][10]) of a geometry, from the real cairo, verified against its own masks."""
    w, h, sx, sy, vx, vy = surface(geom)
    # one glyph at a whole-pixel origin well inside a scratch surface -> its bitmap relative to the origin
    ox, oy, S = 40, 60, 100
    bm = {}
    for c in CHARS:
        out = np.zeros((S, S), np.uint8)
        assert P.cp_text_at(c.encode(), S, S, sx, sy, vx, vy, float(ox), float(oy), out.ctypes.data_as(C.c_void_p)) == 0
        bm[c] = out
    ink = np.maximum.reduce(list(bm.values()))
    ys, xs = np.nonzero(ink)
    left, right, top, bot = xs.min() - ox, xs.max() + 1 - ox, ys.min() - oy, ys.max() + 1 - oy  # the common box, relative to an origin
    gw, gh = int(right - left), int(bot - top)
    alpha = np.stack([bm[c][oy + top:oy + bot, ox + left:ox + right] for c in CHARS])
    # the hinted advance, in pixels
    _, ext = score_mask(P, geom, "0000000")
    adv = ext[4] / 7 * sx
    assert abs(adv - round(adv)) < 1e-9, ("advance is not a whole number of pixels", geom, adv)
    adv = int(round(adv))

    def compose(text, X0, Y0):
        m = np.zeros((h, w), np.int32)
        for i, c in enumerate(text):
            x, y = X0 + i * adv, Y0
            for r in range(gh):
                for q in range(gw):
                    a = int(alpha[CHARS.index(c), r, q])
                    if a and 0 <= y + r < h and 0 <= x + q < w:
                        m[y + r, x + q] = min(255, m[y + r, x + q] + a)  # (boxes may overlap; ink does not, asserted below)
        return m.astype(np.uint8)

    # placement: origin_user = (cx - width / 2, cy + height / 2) -> device -> rounded; width depends on first / last glyph
    x0 = np.zeros((11, 10), np.int16)
    y0s = set()
    rng = np.random.default_rng(5)
    for fi, f in enumerate(CHARS):
        for li, l in enumerate(CHARS[:10]):
            mid = "".join(rng.choice(list(CHARS[:10]), 5))
            text = f + mid + l
            m, ext = score_mask(P, geom, text)
            # (the origin lands on .5 in some geometries, where the last bit of cairo's own matrix arithmetic decides: both
            #  neighbours are tried and cairo's mask says which -- the table is data, not a formula)
            fx, fy = (CX - ext[2] / 2.0 - vx) * sx, (CY + ext[3] / 2.0 - vy) * sy
            cands = [(int(np.floor(fx + 0.5)) + dx + int(left), int(np.floor(fy + 0.5)) + dy + int(top)) for dx in (0, -1, 1) for dy in (0, -1, 1)]
            hits = [(X, Y) for X, Y in cands if np.array_equal(compose(text, X, Y), m)]
            assert len(hits) == 1, ("atlas + rule != cairo's mask", geom, text, fx, fy, hits)
            X0, Y0 = hits[0]
            x0[fi, li] = X0
            y0s.add(Y0)
    assert len(y0s) == 1, y0s
    y0 = y0s.pop()
    for _ in range(300):  # and random strings
        v = int(rng.integers(0, 10 ** 7)) if rng.random() < .8 else -int(rng.integers(0, 10 ** 6))
        text = "%07d" % v
        m, _ = score_mask(P, geom, text)
        assert np.array_equal(compose(text, int(x0[CHARS.index(text[0]), int(text[-1])]), y0), m), (geom, text)
    return alpha, np.array([gw, gh, adv, y0], np.int32), x0


def text_over(frame, points, alpha, layout, x0):
    """grey .5 (128) through the glyphs' coverage OVER the frame, pixman's arithmetic (the model the tests share: oracle/render_np.py)"""
    return R.score_text_atlas(frame, points, dict(alpha=alpha, layout=layout, x0=x0))


def main():
    P = probe()
    geoms = [DEFAULT] + [(float(g[0]), tuple(int(v) for v in g[1:5]), float(g[5])) for g in np.load(os.path.join(HERE, "geometries.npz"))["geometries"]]
    out = {}
    for k, geom in enumerate(geoms):
        alpha, layout, x0 = atlas_of(P, geom)
        out["alpha_%d" % k], out["layout_%d" % k], out["x0_%d" % k] = alpha, layout, x0
        out["geometry_%d" % k] = np.array([geom[0], *geom[1], geom[2]], np.float64)
        print("geometry %d %s: glyph box %d x %d, advance %d, rows %d..%d, first column %d..%d" % (
            k, geom, layout[0], layout[1], layout[2], layout[3], layout[3] + layout[1] - 1, x0.min(), x0.max()))
    g = O.RefDrawGame("youturn")
    meta = dict(cairo=g.cairo_version(), chars=CHARS, grey=128, operator="pixman OVER: mul_un8(128, a) + mul_un8(d, 255 - a)",
                font="cairo toy font 'monospace' bold, 30 user units -> fontconfig -> DejaVu Sans Mono Bold -> FreeType A8, hint metrics on",
                source="alpha: the image's libcairo 1.16 through oracle/cairo_probe.c (cp_text_at); placement checked against cp_score_mask; "
                       "everything checked against oracle/_ref/libsfrefdraw.so (SRC/draw.cpp) frames: scores.npz")
    out["meta"] = json.dumps(meta)

    # ---- frames of the reference's own renderer ------------------------------------------------------------------------------
    A0 = (out["alpha_0"], out["layout_0"], out["x0_0"])
    hexp = g.hex_points()
    rng = np.random.default_rng(20261006)
    b = g.snapshot().copy()
    b["missile_alive"][:] = 0
    b["shell_alive"][:] = 0
    quiet = None
    pts = [0, 1, 9, 10, 99, 100, 511, 512, 1234567, 9999999, 7000001, 1000007, -1, -5, -512, -999999]
    pts += [int(f) * 10 ** 6 + int(rng.integers(0, 10 ** 5)) * 10 + l for f in range(10) for l in range(10)]  # every first / last pair
    pts += [-(int(rng.integers(0, 10 ** 5)) * 10 + l) for l in range(10)]
    while len(pts) < 2200:
        e = int(rng.integers(1, 8))
        pts.append(int(rng.integers(0, 10 ** e)))
    rows = []
    for p in pts:
        s = b.copy()
        s["points"] = p
        g.load_snapshot(s)
        f = g.draw()
        if quiet is None:
            quiet = R.render_raw(s, hexp[:12], hexp[12:], text=False)
        assert np.array_equal(text_over(quiet, p, *A0), f), ("atlas != the reference's frame", p)
        rows.append(f[:9].copy())
    frames, snaps = [], []
    for k in range(320):  # objects under the text: what the glyphs are composited OVER
        s = b.copy()
        s["points"] = int(rng.integers(0, 10 ** int(rng.integers(1, 8))))
        if k % 3 == 0:
            s["ship_alive"] = 0
            s["ship_x"], s["ship_y"] = rng.uniform(270, 440), rng.uniform(84, 135)
        else:
            s["ship_x"], s["ship_y"], s["ship_angle"] = rng.uniform(285, 425), rng.uniform(88, 125), int(rng.integers(0, 360))
        for i in rng.permutation(20)[:int(rng.integers(0, 9))]:
            s["missile_alive"][i] = 1
            s["missile_x"][i], s["missile_y"][i] = rng.uniform(280, 430), rng.uniform(82, 112)
            s["missile_angle"][i] = int(rng.integers(0, 360))
        for i in rng.permutation(20)[:int(rng.integers(0, 3))]:
            s["shell_alive"][i] = 1
            s["shell_x"][i], s["shell_y"][i] = rng.uniform(280, 430), rng.uniform(84, 110)
            a = int(rng.integers(0, 360)) + rng.uniform(.05, .95)
            s["shell_angle"][i] = a
            s["shell_vx"][i], s["shell_vy"][i] = 6.0 * np.cos(np.radians(a)), 6.0 * np.sin(np.radians(a))
        s["vlner"] = int(rng.integers(0, 13))
        g.load_snapshot(s)
        f = g.draw()
        under = R.render_raw(s, hexp[:12], hexp[12:], text=False)
        assert np.array_equal(text_over(under, s["points"], *A0), f), ("atlas OVER objects != the reference's frame", k)
        frames.append(f)
        snaps.append(s)
    np.savez_compressed(os.path.join(HERE, "score_glyphs.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "scores.npz"), rows=np.stack(rows), points=np.array(pts, np.int32), base=np.array(b, O.SNAPSHOT_DTYPE),
                        frames=np.stack(frames), snaps=np.array(snaps, O.SNAPSHOT_DTYPE), hex_points=hexp,
                        meta=json.dumps(dict(meta, surface=[92, 90], rows="rows 0..8 of the default surface for `points` on `base`")))
    print("scores: %d scores' rows, %d frames with objects under the text -- all equal to the atlas model" % (len(rows), len(frames)))


if __name__ == "__main__":
    main()
