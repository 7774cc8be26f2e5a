"""Cost of each phase of the render kernel, by removal: builds libsfmi variants with SF_RENDER_SKIP
bits into build/abl/ (`--build`, run in the build container) and times them (`--run`, on the GPU)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"full": 0, "no_shipfort": 1, "no_projectiles": 2, "no_score": 4, "no_bar": 8, "no_resample": 16,
            "only_copy": 31, "only_resample": 15, "no_cover": 32, "no_composite": 64, "no_cover_composite_resample": 112, "no_live_ship": 128, "no_dead_ship": 256, "no_fortress": 512, "miss_as_hit": 1024}
ONLY = [a for a in sys.argv[1:] if a in VARIANTS]
if ONLY:
    VARIANTS = {k: VARIANTS[k] for k in ONLY}
if "--build" in sys.argv:
    from spacefortress_amd import build as B
    os.makedirs(os.path.join(ROOT, "build/abl"), exist_ok=True)
    for name, bits in VARIANTS.items():
        out = os.path.join(ROOT, "build/abl/libsfmi_render_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DSF_RENDER_SKIP=%d" % bits] + \
              [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out]
        subprocess.check_call(cmd)
        print("built", out)
if "--run" in sys.argv:
    for name in VARIANTS:
        env = dict(os.environ, SFMI_LIB_PATH=os.path.join(ROOT, "build/abl/libsfmi_render_%s.so" % name))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/image_probe.py"), "16384", "100", "image"], env=env,
                           capture_output=True, text=True)
        print("%-16s %s" % (name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
