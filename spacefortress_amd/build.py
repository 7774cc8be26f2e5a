"""Build recipe for libsfmi.so (HIP kernels + C ABI), in-tree, for gfx950 only.

    python -m spacefortress_amd.build

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the contract:
the reference engine rounds a*b+c twice (baseline x86-64 build, no FMA) and the
kernels must do the same to stay bit-exact.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsfmi.so")
SOURCES = ["sf_kernels.hip", "sf_render.hip", "sf_render_generic.hip", "sf_normalize.hip", "sf_rollout_ops.hip", "sf_capi.cpp", "sf_norm_capi.cpp", "sf_host.cpp", "sf_image.cpp"]
HEADERS = ["sf_layout.h", "sf_drawrec.h", "sf_cover.h", "sf_internal.h", "sf_raster.h", "sf_render_tables.h", "sf_deg_dd.h", os.path.join(ROOT, "include", "sfmi.h")]


# -amdgpu-kernarg-preload-count: the first eight kernel parameters (as many as fit the 14 free user SGPRs) arrive in
# SGPRs at wave launch (gfx950 command processor) instead of through a scalar-load round trip to the
# kernel-argument segment; sf_step_kernel's leading parameters are ordered for it: what round trip 1 needs first
# (9.43 -> 9.18 us per launch at 65 536 envs), then the reward / done / info pointers of the epilogue (8.76 -> 8.69).
# A firmware without the feature runs the compiler's compatibility preamble, which loads them the old way.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-amdgpu-kernarg-preload-count=8",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def source_hash(extra_flags=()):
    """sha256 (16 hex digits) over every source, header and compiler flag that goes into libsfmi.so.  It is compiled
    into the library (-DSFMI_BUILD_ID, returned by sf_build_id()), so a test on the GPU box can assert that the binary
    which travelled there was built from the sources which travelled with it."""
    import hashlib

    h = hashlib.sha256()
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [x if os.path.isabs(x) else os.path.join(CSRC, x) for x in HEADERS]
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
    h.update(" ".join(f for f in FLAGS + list(extra_flags) if not f.startswith("-I")).encode())
    return h.hexdigest()[:16]


def built_id(path=None):
    """The build id compiled into an existing libsfmi.so (None if it has none / does not load)."""
    import ctypes

    path = path or LIB
    try:
        L = ctypes.CDLL(path)
        L.sf_build_id.restype = ctypes.c_char_p
        return L.sf_build_id().decode()
    except Exception:
        return None


def needs_build():
    if not os.path.exists(LIB):
        return True
    idf = LIB + ".id"  # sidecar written by build(): spares loading the library just to ask
    have = open(idf).read().strip() if os.path.exists(idf) and os.path.getmtime(idf) >= os.path.getmtime(LIB) else built_id()
    return have != source_hash()


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bid = source_hash(extra_flags)
    cmd = [hipcc] + FLAGS + list(extra_flags) + ['-DSFMI_BUILD_ID="%s"' % bid] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(LIB + ".id", "w") as f:
        f.write(bid + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
