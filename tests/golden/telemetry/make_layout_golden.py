"""Measures the layout of the reference's own rendering in its documentation screenshot
(/root/reference/rl/imgs/screens.png, middle panel: the clean one) and writes the numbers -- not the picture --
to tests/golden/telemetry/screens_layout.json: extents of the big hexagon, the score text and the vulnerability
bar in panel pixels, and the grey levels of text and bar.  Build container only.

    python tests/golden/telemetry/make_layout_golden.py
"""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
img = np.asarray(Image.open("/root/reference/rl/imgs/screens.png").convert("RGB")).astype(int)
out = {}
for name, (x0, x1) in (("left", (0, 455)), ("middle", (476, 931)), ("right", (952, 1407))):
    p = img[:, x0:x1]
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    green = (g > 150) & (r < 100) & (b < 100)
    grey = (abs(r - g) < 12) & (abs(g - b) < 12) & (r > 40)
    grey[:, :3] = grey[:, -3:] = False  # panel frame
    grey[:3] = grey[-3:] = False
    ys, xs = np.nonzero(green)
    ty, tx = np.nonzero(grey[:45, 100:360])
    by, bx = np.nonzero(grey[430:462, 60:400])
    bar_levels = r[430:462, 60:400][grey[430:462, 60:400]]
    out[name] = dict(
        hex_x=[int(xs.min()), int(xs.max())], hex_y=[int(ys.min()), int(ys.max())],
        text_x=[int(tx.min()) + 100, int(tx.max()) + 100], text_y=[int(ty.min()), int(ty.max())],
        text_grey_max=int(r[:45, 100:360][grey[:45, 100:360]].max()),
        bar_x=[int(bx.min()) + 60, int(bx.max()) + 60], bar_y=[int(by.min()) + 430, int(by.max()) + 430],
        bar_grey_mode=int(np.bincount(bar_levels).argmax()))
json.dump(out, open(os.path.join(HERE, "screens_layout.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# ---- the right panel shows a fortress explosion (drawExplosion, SRC/draw.cpp:145-175) around (355, 315): polar
# coordinates of its yellow / red pixels relative to the small hexagon's centre -> explosion_pixels.npz
p = img[:, 952:1407]
r, g, b = p[..., 0], p[..., 1], p[..., 2]
green = (g > 150) & (r < 100) & (b < 100)
y2, x2 = np.nonzero(green[150:330, 150:310])
cx, cy = (x2.min() + x2.max()) / 2 + 150 + 0.5, (y2.min() + y2.max()) / 2 + 150 + 0.5
cols = {"yellow": (r > 150) & (g > 150) & (b < 100), "red": (r > 150) & (g < 100) & (b < 100)}
arr = {}
for name, m in cols.items():
    yy, xx = np.nonzero(m)
    rr = np.hypot(xx + 0.5 - cx, yy + 0.5 - cy)
    th = np.degrees(np.arctan2(yy + 0.5 - cy, xx + 0.5 - cx)) % 360
    keep = rr < 75  # the ship (also yellow) is farther out
    arr[name + "_r"] = rr[keep].astype(np.float32)
    arr[name + "_theta"] = th[keep].astype(np.float32)
np.savez_compressed(os.path.join(HERE, "explosion_pixels.npz"), **arr)
print({k: v.shape for k, v in arr.items()})

# ---- the left and middle panels show a live fortress (wireframe of SRC/wireframe.cpp:53-67 at (355, 315)): its
# yellow pixels, in user units relative to the small hexagon's centre -> fortress_pixels.npz
fort = {}
for name, (x0, x1) in (("left", (0, 455)), ("middle", (476, 931))):
    p = img[:, x0:x1]
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    green = (g > 150) & (r < 100) & (b < 100)
    y2, x2 = np.nonzero(green[150:330, 150:310])
    cx, cy = (x2.min() + x2.max()) / 2 + 150 + 0.5, (y2.min() + y2.max()) / 2 + 150 + 0.5
    yy, xx = np.nonzero((r > 150) & (g > 150) & (b < 100))
    dx, dy = xx + 0.5 - cx, yy + 0.5 - cy
    keep = np.hypot(dx, dy) < 45
    fort[name] = np.stack([dx[keep], dy[keep]], 1).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "fortress_pixels.npz"), **fort)
print({k: v.shape for k, v in fort.items()})

# ---- every panel shows a live ship (wireframe of SRC/wireframe.cpp:39-51, position and heading unknown): its
# yellow pixels in panel coordinates -> ship_pixels.npz
ship = {}
for name, (x0, x1) in (("left", (0, 455)), ("middle", (476, 931)), ("right", (952, 1407))):
    p = img[:, x0:x1]
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    green = (g > 150) & (r < 100) & (b < 100)
    y2, x2 = np.nonzero(green[150:330, 150:310])
    cx, cy = (x2.min() + x2.max()) / 2 + 150 + 0.5, (y2.min() + y2.max()) / 2 + 150 + 0.5
    yy, xx = np.nonzero((r > 150) & (g > 150) & (b < 100))
    keep = np.hypot(xx + 0.5 - cx, yy + 0.5 - cy) > 75
    ship[name] = np.stack([xx[keep] + 0.5, yy[keep] + 0.5], 1).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "ship_pixels.npz"), **ship)
print({k: v.shape for k, v in ship.items()})
