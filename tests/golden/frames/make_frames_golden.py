#!/usr/bin/env python3
"""Frame fixtures from the reference's REAL renderer.

Run in the build container only: needs /root/reference and the image's cairo 1.16 (`make -C oracle refdraw` builds
oracle/_ref/libsfrefdraw.so = the reference's engine, draw.cpp and wireframe.cpp, compiled where they lie, plus our driver).
What is written is data -- engine states and the pixels the reference drew for them -- never reference source, and the
library itself stays behind (oracle/_ref/ is git-ignored).

    python tests/golden/frames/make_frames_golden.py

Every frame is what SSF_Env._draw gets (ENV:203-206): Game.draw() on a Game(gametype, ls, grayscale=True, w, h, viewport)
-> drawGameStateScaled (SRC/draw.cpp:256-270) -> one channel of the RGB24 surface (R = G = B in grayscale mode).

    scenarios.npz   every golden scenario of tests/golden/*.npz at a stride: the frame of the state after step t
    poses.npz       constructed states loaded into the reference Game: the ship at every heading 0..359 (at sub-pixel
                    offsets), the fortress at its 36 headings, ship / fortress explosions, every bar state incl. kill-ready,
                    shells inside / outside the 21-unit rule (SRC/draw.cpp:248-252), crowded pools (20 missiles + 20
                    shells), objects across the surface's borders, scores
    geometries.npz  other SSF_Env(scale, viewport, ls) geometries incl. one whose vw * scale is not whole
                    (scale_x = w / vw != scale: SRC/draw.cpp:70-71)
each with   frames u8[n, H, W] (geometries: one array per geometry), snaps sfo_snapshot[n], gametype index, label,
            hex_points f64[24] and meta json (cairo version, the font note, the masked text rows).

The score text is drawn by cairo's toy font API ("monospace" bold 30, SRC/draw.cpp:161-173), i.e. by whatever font
fontconfig resolves on the box (here DejaVu Sans Mono Bold): rows 0..8 of the default surface are box-dependent and every
consumer of these files masks them (meta["text_rows"]).
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
ROOT = os.path.dirname(os.path.dirname(GOLDEN))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

GAMETYPES = ["youturn", "autoturn", "test-youturn", "test-autoturn"]
FORT = (355.0, 315.0)


def set_shell_heading(s, i, a):
    """A shell's heading with the velocity that has it (the engine keeps both, SRC/game.cpp:159-173; a batch keeps the velocity
    and derives the drawn heading): away from whole degrees, where float noise could truncate to the neighbour."""
    s["shell_angle"][i] = a
    s["shell_vx"][i], s["shell_vy"][i] = 6.0 * np.cos(np.radians(a)), 6.0 * np.sin(np.radians(a))


def base_snap(g):
    s = g.snapshot().copy()
    s["missile_alive"][:] = 0
    s["shell_alive"][:] = 0
    return s


def main():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "refdraw"], stdout=subprocess.DEVNULL)
    games = {gt: O.RefDrawGame(gt) for gt in GAMETYPES}
    g0 = games["youturn"]
    meta = dict(cairo=g0.cairo_version(), text_rows=9, surface=[92, 90],
                font="score text: cairo toy font 'monospace' bold 30 -> fontconfig -> DejaVu Sans Mono Bold on this image; "
                     "rows 0..8 are box-dependent and masked",
                source="oracle/_ref/libsfrefdraw.so: SRC/{game,vector,object,hexagon,config,configs,wireframe,draw}.cpp")
    hexp = g0.hex_points()

    # ---- scenarios ---------------------------------------------------------------------------------------------------------
    frames, snaps, gti, labels = [], [], [], []
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        name = os.path.basename(path)[:-4]
        if name == "tables":
            continue
        z = np.load(path)
        m = json.loads(str(z["meta"]))
        g = games[m["gametype"]]
        assert np.array_equal(g.hex_points(), hexp)
        S = z["snaps"]
        # dense where something happens (explosions, shells, many missiles), sparse elsewhere
        busy = (S["ship_alive"] == 0) | (S["fort_alive"] == 0) | (S["shell_alive"].sum(1) > 0) | (S["missile_alive"].sum(1) > 2)
        idx = sorted(set(range(0, len(S), max(1, len(S) // 25))) | set(np.flatnonzero(busy)[::max(1, busy.sum() // 25)]))
        for t in idx:
            g.load_snapshot(S[t])
            frames.append(g.draw())
            snaps.append(S[t])
            gti.append(GAMETYPES.index(m["gametype"]))
            labels.append("%s@%d" % (name, t * m["snap_every"]))
    np.savez_compressed(os.path.join(HERE, "scenarios.npz"), frames=np.stack(frames), snaps=np.array(snaps, O.SNAPSHOT_DTYPE),
                        gametype=np.array(gti, np.int8), labels=np.array(labels), hex_points=hexp, meta=json.dumps(meta))
    print("scenarios: %d frames" % len(frames))

    # ---- poses ---------------------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(20261005)
    frames, snaps, labels = [], [], []
    g = g0
    b = base_snap(g)

    def emit(s, label):
        g.load_snapshot(s)
        frames.append(g.draw())
        snaps.append(s.copy())
        labels.append(label)

    for a in range(360):  # the ship at every heading, off the pixel grid
        s = b.copy()
        s["ship_alive"], s["ship_angle"] = 1, a
        s["ship_x"], s["ship_y"] = 355 + 120 * np.cos(a * 0.7) + rng.uniform(-3, 3), 315 + 110 * np.sin(a * 1.3) + rng.uniform(-3, 3)
        s["fort_angle"] = (a // 10) * 10 % 360
        emit(s, "ship_heading_%d" % a)
    for a in range(0, 360, 10):  # the fortress' 36 headings, ship far away / on the grid
        s = b.copy()
        s["ship_x"], s["ship_y"], s["ship_angle"], s["fort_angle"] = 250.0, 200.0, 90, a
        emit(s, "fort_heading_%d" % a)
    for k in range(24):  # ship explosions anywhere (incl. whole-pixel and near-border places), fortress alive
        s = b.copy()
        s["ship_alive"] = 0
        s["ship_x"], s["ship_y"] = (rng.uniform(160, 550), rng.uniform(110, 510)) if k % 4 else (float(rng.integers(160, 550)), float(rng.integers(110, 510)))
        emit(s, "ship_explosion_%d" % k)
    for k in range(6):  # fortress explosion, ship alive / dead next to it
        s = b.copy()
        s["fort_alive"] = 0
        s["ship_alive"] = k % 2
        s["ship_x"], s["ship_y"], s["ship_angle"] = 355 + 30 * k, 315 - 20 * k, 45 * k
        emit(s, "fort_explosion_%d" % k)
    for v in range(0, 14):  # the bar: every state, kill-ready or not
        for timer in (0, 249, 250, 1000):
            s = b.copy()
            s["vlner"], s["fort_vuln_timer"] = v, timer
            emit(s, "bar_%d_%d" % (v, timer))
    for k, d in enumerate((0.0, 10.0, 20.9, 21.0, 21.000001, 21.5, 25.0, 60.0, 150.0)):  # the 21-unit rule
        for a in (0.4, 33.7, 180.3, 271.2, 359.9):
            s = b.copy()
            s["shell_alive"][3] = 1
            s["shell_x"][3], s["shell_y"][3] = FORT[0] + d * np.cos(np.radians(a)), FORT[1] + d * np.sin(np.radians(a))
            set_shell_heading(s, 3, a)  # a double: drawWireFrame's int parameter truncates it
            emit(s, "shell_rule_%d_%g" % (k, a))
    for k in range(40):  # crowded pools
        s = b.copy()
        nm, ns = (20, 20) if k < 8 else (int(rng.integers(1, 21)), int(rng.integers(0, 21)))
        for i in rng.permutation(20)[:nm]:
            s["missile_alive"][i] = 1
            s["missile_x"][i], s["missile_y"][i] = rng.uniform(100, 620), rng.uniform(50, 570)
            s["missile_angle"][i] = int(rng.integers(0, 360))
        for i in rng.permutation(20)[:ns]:
            s["shell_alive"][i] = 1
            s["shell_x"][i], s["shell_y"][i] = rng.uniform(100, 620), rng.uniform(50, 570)
            set_shell_heading(s, i, int(rng.integers(0, 360)) + rng.uniform(.05, .95))
        s["ship_x"], s["ship_y"], s["ship_angle"] = rng.uniform(200, 500), rng.uniform(150, 480), int(rng.integers(0, 360))
        s["ship_alive"] = int(k % 5 != 0)
        s["fort_alive"] = int(k % 7 != 0)
        s["fort_angle"] = int(rng.integers(0, 36)) * 10
        s["vlner"] = int(rng.integers(0, 13))
        emit(s, "crowd_%d" % k)
    for k in range(60):  # objects across the four borders of the view (130..580 x 80..540) and near the HUD
        s = b.copy()
        side = k % 4
        t = rng.uniform(0, 1)
        x, y = [(130 + rng.uniform(-6, 6), 80 + 460 * t), (580 + rng.uniform(-6, 6), 80 + 460 * t),
                (130 + 450 * t, 80 + rng.uniform(-6, 6)), (130 + 450 * t, 540 + rng.uniform(-6, 6))][side]
        i = int(rng.integers(0, 20))
        if k % 3 == 2:
            s["shell_alive"][i], s["shell_x"][i], s["shell_y"][i] = 1, x, y
            set_shell_heading(s, i, int(rng.integers(0, 360)) + rng.uniform(.05, .95))
        else:
            s["missile_alive"][i], s["missile_x"][i], s["missile_y"][i], s["missile_angle"][i] = 1, x, y, int(rng.integers(0, 360))
        j = (i + 7) % 20  # ... and one over the score text / the bar
        s["missile_alive"][j], s["missile_angle"][j] = 1, int(rng.integers(0, 360))
        s["missile_x"][j], s["missile_y"][j] = (rng.uniform(290, 420), rng.uniform(85, 112)) if k % 2 else (rng.uniform(250, 460), rng.uniform(515, 538))
        emit(s, "border_%d" % k)
    for k, pts in enumerate((0, 7, -3, 123, 1234567, -99999, 511, -512)):  # scores (text rows: masked, kept for the record)
        s = b.copy()
        s["points"] = pts
        emit(s, "score_%d" % pts)
    np.savez_compressed(os.path.join(HERE, "poses.npz"), frames=np.stack(frames), snaps=np.array(snaps, O.SNAPSHOT_DTYPE),
                        gametype=np.zeros(len(frames), np.int8), labels=np.array(labels), hex_points=hexp, meta=json.dumps(meta))
    print("poses: %d frames" % len(frames))

    # ---- other geometries ----------------------------------------------------------------------------------------------------
    geoms = [(.25, (100, 60, 500, 520), 2.0), (.3, (130, 80, 450, 460), 4.5), (.25, (130, 80, 450, 460), 3.0),
             (.4, (130, 80, 450, 460), 3.0)]
    out = dict(hex_points=hexp, snaps=None)
    z = np.load(os.path.join(GOLDEN, "youturn_hunter.npz"))
    S = z["snaps"][::max(1, len(z["snaps"]) // 14)][:14]
    extra = []
    for k in range(6):
        s = b.copy()
        s["ship_alive"] = int(k % 2)
        s["fort_alive"] = int(k % 3 != 0)
        s["fort_angle"] = 10 * int(rng.integers(0, 36))
        s["ship_x"], s["ship_y"], s["ship_angle"] = rng.uniform(200, 500), rng.uniform(150, 480), int(rng.integers(0, 360))
        s["vlner"] = 3 * k
        for i in range(3):
            s["missile_alive"][i], s["missile_x"][i], s["missile_y"][i], s["missile_angle"][i] = 1, rng.uniform(110, 600), rng.uniform(70, 560), int(rng.integers(0, 360))
        s["shell_alive"][0], s["shell_x"][0], s["shell_y"][0] = 1, rng.uniform(150, 550), rng.uniform(100, 500)
        set_shell_heading(s, 0, int(rng.integers(0, 360)) + rng.uniform(.05, .95))
        extra.append(s)
    S = np.concatenate([S, np.array(extra, O.SNAPSHOT_DTYPE)])
    out["snaps"] = S
    out["geometries"] = np.array([[sc, *vp, ls] for sc, vp, ls in geoms], np.float64)
    for gi, (sc, vp, ls) in enumerate(geoms):
        fr = []
        for s in S:
            g.load_snapshot(s)
            fr.append(g.draw(scale=sc, viewport=vp, ls=ls))
        out["frames_%d" % gi] = np.stack(fr)
    out["meta"] = json.dumps(dict(meta, text_rows="int(112 * scale_y) + 1 rows from the top (user y <= 112)"))
    np.savez_compressed(os.path.join(HERE, "geometries.npz"), **out)
    print("geometries: %d x %d frames" % (len(geoms), len(S)))


if __name__ == "__main__":
    main()
