#!/usr/bin/env python3
"""A/B timing of the image observation for differently built libsfmi variants on ONE device, interleaved rounds:
    python tools/ab_render.py build/abl/libsfmi_a.so build/abl/libsfmi_b.so [--rounds 3] [--envs 16384] [--steps 300]
Each variant runs tools/image_probe.py in its own subprocess per round; prints the medians of `sf_step + render` and of the
render launch alone (us per 16 384 frames)."""
import argparse, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--envs", type=int, default=16384)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--mode", default="image")
a = ap.parse_args()
res = {l: [] for l in a.libs}
for r in range(a.rounds):
    for l in a.libs:
        env = dict(os.environ, SFMI_LIB_PATH=os.path.abspath(l))
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "image_probe.py"), str(a.envs), str(a.steps)] + a.mode.split(),
                                      env=env, stderr=subprocess.DEVNULL, text=True)
        m = re.search(r"step\+render ([0-9.]+) us .*render alone ([0-9.]+) us.*?(?:step alone ([0-9.]+) us)?$", out, re.M)
        res[l].append((float(m.group(1)), float(m.group(2)), float(m.group(3) or 0)))
for l, v in res.items():
    sr = sorted(x[0] for x in v); ra = sorted(x[1] for x in v); sa = sorted(x[2] for x in v)
    print("%-36s step+render median %.1f min %.1f | render alone median %.1f min %.1f | step alone median %.2f  (%s)" % (
        os.path.basename(l), sr[len(sr) // 2], sr[0], ra[len(ra) // 2], ra[0], sa[len(sa) // 2], " ".join("%.1f/%.1f/%.2f" % x for x in v)), flush=True)
