#!/usr/bin/env python3
"""Build libsfmi variants that differ in ONE source's -D switches, reusing cached objects of the others:
    python tools/variant_fast.py sf_render.hip NAME1:-DX=1,-DY=2 NAME2:-DX=3 ...   ->  build/abl/libsfmi_NAME.so
(objects of the unchanged sources are compiled once into build/obj/, keyed by the library's source hash)."""
import concurrent.futures, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spacefortress_amd import build as B
hipcc = "/opt/rocm/bin/hipcc"
bid = B.source_hash()
objdir = os.path.join(ROOT, "build", "obj", bid)
os.makedirs(objdir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "build", "abl"), exist_ok=True)
cflags = [f for f in B.FLAGS if f != "-shared"] + ['-DSFMI_BUILD_ID="%s"' % bid]
target = sys.argv[1]
jobs = []
for s in B.SOURCES:
    if s != target:
        o = os.path.join(objdir, s + ".o")
        if not os.path.exists(o):
            jobs.append((s, o, []))
variants = []
for spec in sys.argv[2:]:
    name, _, flags = spec.partition(":")
    fl = [f for f in flags.split(",") if f]
    o = os.path.join(objdir, "%s.%s.o" % (target, name))
    jobs.append((target, o, fl))
    variants.append((name, o))
def one(job):
    s, o, fl = job
    subprocess.check_call([hipcc] + cflags + fl + ["-x", "hip", "-c", os.path.join(B.CSRC, s), "-o", o])
with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
    list(ex.map(one, jobs))
for name, o in variants:
    out = os.path.join(ROOT, "build", "abl", "libsfmi_%s.so" % name)
    objs = [o if s == target else os.path.join(objdir, s + ".o") for s in B.SOURCES]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "--hip-link"] + objs + ["-o", out])
    print(out)
