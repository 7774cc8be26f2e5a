#!/usr/bin/env python3
"""Build a libsfmi variant with extra -D switches into build/abl/ (for tools/ab.py):
    python tools/variant.py NAME [-DSF_TRIG_HOIST=0 ...]   ->  build/abl/libsfmi_NAME.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spacefortress_amd import build as B
out = os.path.join(ROOT, "build", "abl", "libsfmi_%s.so" % sys.argv[1])
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + sys.argv[2:] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out])
print(out)
