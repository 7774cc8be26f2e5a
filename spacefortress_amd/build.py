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
SOURCES = ["sf_kernels.hip", "sf_render.hip", "sf_normalize.hip", "sf_rollout_ops.hip", "sf_capi.cpp", "sf_norm_capi.cpp", "sf_host.cpp", "sf_image.cpp"]
HEADERS = ["sf_layout.h", "sf_internal.h", "sf_raster.h", "sf_render_tables.h", "sf_deg_dd.h", os.path.join(ROOT, "include", "sfmi.h")]


# -amdgpu-kernarg-preload-count: the first eight kernel parameters (as many as fit the 14 free user SGPRs) arrive in
# SGPRs at wave launch (gfx950 command processor) instead of through a scalar-load round trip to the
# kernel-argument segment; sf_step_kernel's leading parameters are ordered for it: what round trip 1 needs first
# (9.43 -> 9.18 us per launch at 65 536 envs), then the reward / done / info pointers of the epilogue (8.76 -> 8.69).
# A firmware without the feature runs the compiler's compatibility preamble, which loads them the old way.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-amdgpu-kernarg-preload-count=8",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
