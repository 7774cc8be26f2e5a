"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the image observation.

What the reference draws, where, in which order and in which grey comes from its renderer
(SRC = /root/reference/python/spacefortress/src):

  drawGameStateScaled   SRC/draw.cpp:257-270   scale .2, translate (-130,-80), black paint
  drawJustGameStuff     SRC/draw.cpp:227-255   hexagons, ship|explosion, fortress|explosion,
                                               missiles, shells (only when > 21 from the fortress)
  drawWireFrame         SRC/draw.cpp:112-129   translate, rotate(int angle), lines, one stroke
  wireframes            SRC/wireframe.cpp:11-67
  drawExplosion         SRC/draw.cpp:145-175
  drawScore             SRC/draw.cpp:190-203   "%07d", grey .5
  drawVlner             SRC/draw.cpp:205-225
  SSF_Env._draw         ENV:203-206            grey channel of the RGB24 surface, [92][90] uint8
  WrapPyTorch           rl/envs.py:28-30       cv2.resize(.., (84, 84), INTER_AREA)

PARITY UNPINNED at the pixel level: cairo, freetype and cv2 are not in this image and the
reference ships no frame fixtures.  Its documentation screenshot rl/imgs/screens.png (full colour, scale 1)
pins the layout of the score text and the bar and the grey levels 128 / 84 / 168
(tests/golden/telemetry/screens_layout.json); how a 0.6-pixel stroke turns into grey levels is a MODEL here -- exact area coverage of
each stroke's rectangles, 8-bit OVER compositing, seven-segment digits for the score text, one chord
per explosion arc and a 12-gon ring for its circle; INTER_AREA follows OpenCV's published algorithm (imgproc/resize.cpp,
computeResizeAreaTab + resizeArea_).  This file restates that model independently of the HIP
kernel (float64, polygon clipping instead of the kernel's edge integrals) so that the kernel can be
checked against it; geometry-level checks (where the ship is, what lights up) are in the tests.
"""
import math

import numpy as np

W, H, OUT = 90, 92, 84
VP_X, VP_Y, SCALE, LINE_W = 130.0, 80.0, 0.2, 3.0
FORT = (355.0, 315.0)


def set_geometry(scale=.2, viewport=(130, 80, 450, 460), ls=3):
    """The geometry SSF_Env(scale, viewport, ls) asks for (ENV:50-60): surface int(vw * scale) x int(vh * scale), device =
    (user - (vx, vy)) * scale, line width ls user units.  Module-wide (the functions below read the globals); returns the
    previous (scale, viewport, ls) so that a test can put it back."""
    global W, H, VP_X, VP_Y, SCALE, LINE_W
    prev = (SCALE, (VP_X, VP_Y, W / SCALE, H / SCALE), LINE_W)
    VP_X, VP_Y, SCALE, LINE_W = float(viewport[0]), float(viewport[1]), float(scale), float(ls)
    W, H = int(viewport[2] * scale), int(viewport[3] * scale)
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


def _dev(pts):
    pts = np.asarray(pts, np.float64)
    return np.stack([(pts[:, 0] - VP_X) * SCALE, (pts[:, 1] - VP_Y) * SCALE], 1)


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
    """Composite convex polygon `poly` (device coordinates) in `grey` OVER the uint8 frame."""
    poly = np.asarray(poly, np.float64)
    x0, y0 = np.floor(poly.min(0)).astype(int)
    x1, y1 = np.ceil(poly.max(0)).astype(int)
    for py in range(max(y0, 0), min(y1, H)):
        for px in range(max(x0, 0), min(x1, W)):
            a = min(pixel_area(poly, px, py), 1.0)
            m = int(a * 255.0 + 0.5)
            if m > 0:
                fb[py, px] = mul_un8(grey, m) + mul_un8(int(fb[py, px]), 255 - m)


def line_poly(line, angle_deg, pos):
    ax, ay, bx, by = line
    ux, uy = bx - ax, by - ay
    ln = math.hypot(ux, uy)
    nx, ny = -uy / ln * LINE_W / 2, ux / ln * LINE_W / 2
    local = [(ax + nx, ay + ny), (bx + nx, by + ny), (bx - nx, by - ny), (ax - nx, ay - ny)]
    c, s = math.cos(math.radians(angle_deg)), math.sin(math.radians(angle_deg))
    return _dev([(pos[0] + c * x - s * y, pos[1] + s * x + c * y) for x, y in local])


def rect_poly(x0, y0, x1, y1):
    return _dev([(x0, y0), (x1, y0), (x1, y1), (x0, y1)])


def wireframe(fb, lines, angle, pos, grey=255):
    for ln in lines:
        over(fb, line_poly(ln, int(angle), pos), grey)


def explosion_arcs():
    """(radius, start degree, end degree, grey) of the 84 arcs of drawExplosion, SRC/draw.cpp:149-166."""
    arcs, ofs = [], 0
    for radius in range(15, 70, 8):
        ofs += 3
        grey = 191 if radius < 60 else 128  # .75 (yellow in colour mode), .5 (red)
        for angle in range(0, 360, 30):
            arcs.append((radius, angle + ofs, angle + ofs + 10, grey))
    return arcs


def explosion(fb, pos):
    for radius, a0, a1, grey in explosion_arcs():
        _arc(fb, pos, radius, a0, a1, grey)
    # the radius-7 circle: ONE stroke (cairo_arc 0..2pi + cairo_stroke), modelled as the ring between
    # two regular 12-gons of circumradius 7 -/+ half the line width
    gons = []
    for r in (7 + LINE_W / 2, 7 - LINE_W / 2):
        gons.append(_dev([(pos[0] + r * math.cos(math.radians(30 * k)), pos[1] + r * math.sin(math.radians(30 * k)))
                          for k in range(12)]))
    x0, y0 = np.floor(gons[0].min(0)).astype(int)
    x1, y1 = np.ceil(gons[0].max(0)).astype(int)
    for py in range(max(y0, 0), min(y1, H)):
        for px in range(max(x0, 0), min(x1, W)):
            a = min(max(pixel_area(gons[0], px, py) - pixel_area(gons[1], px, py), 0.0), 1.0)
            m = int(a * 255.0 + 0.5)
            if m > 0:
                fb[py, px] = mul_un8(191, m) + mul_un8(int(fb[py, px]), 255 - m)


def _arc(fb, pos, r, a0, a1, grey):
    ri, ro = r - LINE_W / 2, r + LINE_W / 2
    c0, s0 = math.cos(math.radians(a0)), math.sin(math.radians(a0))
    c1, s1 = math.cos(math.radians(a1)), math.sin(math.radians(a1))
    over(fb, _dev([(pos[0] + ri * c0, pos[1] + ri * s0), (pos[0] + ro * c0, pos[1] + ro * s0),
                   (pos[0] + ro * c1, pos[1] + ro * s1), (pos[0] + ri * c1, pos[1] + ri * s1)]), grey)


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


_BG = None


def background(hex_big, hex_small):
    """Both hexagons as closed strokes with miter joins: outline polygon minus inline polygon."""
    fb = np.zeros((H, W), np.uint8)
    for pts in (hex_big, hex_small):
        pts = np.asarray(pts, np.float64).reshape(6, 2)
        outer, inner = _offset(pts, LINE_W / 2), _offset(pts, -LINE_W / 2)
        o, i = _dev(outer), _dev(inner)
        for py in range(H):
            for px in range(W):
                a = pixel_area(o, px, py) - pixel_area(i, px, py)
                if a <= 0:
                    continue
                m = int(min(a, 1.0) * 255.0 + 0.5)
                fb[py, px] = mul_un8(255, m) + mul_un8(int(fb[py, px]), 255 - m)
    return fb


def _offset(pts, off):
    """Vertices of the hexagon's edges moved by `off` along their outward normals (miter joins)."""
    px, py = [float(v) for v in pts[:, 0]], [float(v) for v in pts[:, 1]]
    area2 = 0.0
    for i in range(6):
        j = (i + 1) % 6
        area2 += px[i] * py[j] - px[j] * py[i]
    orient = 1.0 if area2 > 0 else -1.0
    nx, ny, c = [], [], []
    for i in range(6):
        j = (i + 1) % 6
        ex, ey = px[j] - px[i], py[j] - py[i]
        ln = math.sqrt(ex * ex + ey * ey)
        nx.append(orient * ey / ln)
        ny.append(-orient * ex / ln)
        c.append(nx[i] * px[i] + ny[i] * py[i] + off)
    out = []
    for i in range(6):
        h = (i + 5) % 6
        det = nx[h] * ny[i] - ny[h] * nx[i]
        out.append(((c[h] * ny[i] - ny[h] * c[i]) / det, (nx[h] * c[i] - c[h] * nx[i]) / det))
    return np.array(out)


def render_raw(snap, hex_big, hex_small, vuln_time=250, bg=None):
    """One [92][90] uint8 frame from an oracle snapshot record (oracle.SNAPSHOT_DTYPE)."""
    fb = (background(hex_big, hex_small) if bg is None else bg).copy()
    ship = (float(snap["ship_x"]), float(snap["ship_y"]))
    if snap["ship_alive"]:
        wireframe(fb, SHIP_LINES, snap["ship_angle"], ship)
    else:
        explosion(fb, ship)
    if snap["fort_alive"]:
        wireframe(fb, FORT_LINES, snap["fort_angle"], FORT)
    else:
        explosion(fb, FORT)
    for i in range(len(snap["missile_alive"])):
        if snap["missile_alive"][i]:
            wireframe(fb, MISSILE_LINES, snap["missile_angle"][i], (snap["missile_x"][i], snap["missile_y"][i]))
    for i in range(len(snap["shell_alive"])):
        if snap["shell_alive"][i]:
            d = math.sqrt((snap["shell_x"][i] - FORT[0]) ** 2 + (snap["shell_y"][i] - FORT[1]) ** 2)
            if d > 21:
                wireframe(fb, SHELL_LINES, snap["shell_angle"][i], (snap["shell_x"][i], snap["shell_y"][i]))
    score_text(fb, snap["points"])
    vlner = int(snap["vlner"])
    kill = vlner > 10 and int(snap["fort_vuln_timer"]) < vuln_time
    over(fb, rect_poly(255, 522, 455, 532), 84)
    if vlner > 0:
        over(fb, rect_poly(255, 522, 255 + 20 * min(vlner, 10), 532), 255 if kill else 168)
    return fb


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
