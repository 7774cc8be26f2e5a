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
