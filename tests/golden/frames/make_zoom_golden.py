#!/usr/bin/env python3
"""One more geometry for tests/golden/frames: a close-up -- SSF_Env(scale=0.75, viewport=(230, 185, 250, 260), ls=3), a 187 x 195
surface around the fortress, the largest scale the renderer takes (sfmi.h: sf_set_image_geometry) -- where a wireframe's box
outgrows one pass of the rasteriser's accumulators, the explosion's arcs are long and the big hexagon lies outside the view.  Same recipe as make_frames_golden.py (the reference's REAL renderer, oracle/_ref/libsfrefdraw.so; build
container only); a file of its own so that the other fixtures stay byte for byte what they were.

    python tests/golden/frames/make_zoom_golden.py        ->  tests/golden/frames/zoom.npz
"""
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
from make_frames_golden import base_snap, set_shell_heading  # noqa: E402

GEOM = (0.75, (230, 185, 250, 260), 3.0)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "refdraw"], stdout=subprocess.DEVNULL)
    g = O.RefDrawGame("youturn")
    rng = np.random.default_rng(77)
    b = base_snap(g)
    snaps = []
    for k in range(24):
        s = b.copy()
        s["ship_alive"] = int(k % 4 != 3)  # every fourth: the ship's explosion (radius up to 63 pixels here)
        s["fort_alive"] = int(k % 5 != 4)  # every fifth: the fortress's
        s["fort_angle"] = 10 * int(rng.integers(0, 36)) if k else 0
        s["ship_x"], s["ship_y"] = rng.uniform(240, 470), rng.uniform(195, 435)
        s["ship_angle"] = int(rng.integers(0, 360)) if k > 1 else 90 * k
        s["vlner"] = k
        s["fort_vuln_timer"] = 100 if k % 2 else 400
        for i in range(int(rng.integers(0, 4))):
            s["missile_alive"][i], s["missile_x"][i], s["missile_y"][i] = 1, rng.uniform(225, 485), rng.uniform(180, 450)
            s["missile_angle"][i] = int(rng.integers(0, 360))
        for i in range(int(rng.integers(0, 3))):
            s["shell_alive"][i], s["shell_x"][i], s["shell_y"][i] = 1, rng.uniform(225, 485), rng.uniform(180, 450)
            set_shell_heading(s, i, int(rng.integers(0, 360)) + rng.uniform(.05, .95))
        snaps.append(s)
    S = np.array(snaps, O.SNAPSHOT_DTYPE)
    frames = []
    for s in S:
        g.load_snapshot(s)
        frames.append(g.draw(scale=GEOM[0], viewport=GEOM[1], ls=GEOM[2]))
    meta = dict(cairo=g.cairo_version(), geometry=[GEOM[0], list(GEOM[1]), GEOM[2]],
                text_rows="none: the score (user y <= 112) and the bar lie outside this viewport's rows 185 .. 445")
    np.savez_compressed(os.path.join(HERE, "zoom.npz"), frames=np.stack(frames), snaps=S, hex_points=g.hex_points(),
                        geometry=np.array([GEOM[0], *GEOM[1], GEOM[2]], np.float64), meta=json.dumps(meta))
    print("zoom: %d frames of %s" % (len(frames), frames[0].shape))


if __name__ == "__main__":
    main()
