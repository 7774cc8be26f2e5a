"""A/B of libsfmi builds on the 4-frame ring path (BASELINE configs[4] as bench.py times it): python tools/stack_ab.py LIB... """
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
res = {l: [] for l in libs}
for r in range(3):
    for l in libs:
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "image_probe.py"), "16384", "400", "stack"],
                                      env=dict(os.environ, SFMI_LIB_PATH=os.path.abspath(l)), stderr=subprocess.DEVNULL, text=True)
        res[l].append(float(re.search(r"ring ([0-9.]+) us", out).group(1)))
for l, v in res.items():
    print("%-28s ring path median %.1f us  (%s)" % (os.path.basename(l), sorted(v)[1], " ".join("%.1f" % x for x in v)))
