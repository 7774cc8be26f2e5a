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
SOURCES = ["sf_kernels.hip", "sf_render.hip", "sf_render_generic.hip", "sf_normalize.hip", "sf_rollout_ops.hip", "sf_capi.cpp", "sf_norm_capi.cpp", "sf_host.cpp", "sf_image.cpp", "sf_cairo_host.cpp"]
HEADERS = ["sf_layout.h", "sf_drawrec.h", "sf_internal.h", "sf_raster.h", "sf_tor.h", "sf_tor_dev.h", "sf_cairo_host.h", "sf_deg_dd.h", "sf_glyphs.h", os.path.join(ROOT, "include", "sfmi.h")]


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


_WIDE_ST = None


def scan_wide_store_hazard(text):
    """gfx950 assembly or disassembly -> [(kernel, store, next instruction)] for every 96- / 128-bit BUFFER store whose
    soffset is an SGPR and whose next instruction is a VALU write of one of its data registers.  The store reads its data
    over several cycles; the compiler inserts the wait state for global / flat stores and for buffer stores with an immediate
    soffset but takes this form to be safe (LLVM GCNHazardRecognizer::createsVALUHazard).  On MI355X it is safe only while the
    wave is alone on its SIMD: with several, lanes 12-15 of every 16 store the NEXT instruction's result (round 4: batches
    beyond 65 536 envs played different games in those lanes).  The kernels' wide buffer stores go out through sf_buf_st128
    (inline assembly, s_nop attached); this is the check that the compiler emitted none of its own."""
    import re

    global _WIDE_ST
    if _WIDE_ST is None:
        _WIDE_ST = (re.compile(r"^\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi)\b"),
                    re.compile(r"^\s*(v_\w+)\s+(?:v\[(\d+):(\d+)\]|v(\d+))(?=[,\s]|$)"),
                    re.compile(r"^(?:[0-9a-f]+ <)?(_Z\w+|sf_\w+)>?:"))
    st, dst, lab = _WIDE_ST
    found, kernel, pending = [], None, None
    for line in text.splitlines():
        m = lab.match(line)
        if m:
            kernel = m.group(1)
            continue
        body = line.strip()
        if not body or body.startswith((";", ".", "//")) or body.endswith(":"):
            continue
        if pending is not None:
            lo, hi, sline = pending
            pending = None
            d = dst.match(line)
            if d and not d.group(1).startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
                dlo, dhi = (int(d.group(2)), int(d.group(3))) if d.group(2) is not None else (int(d.group(4)),) * 2
                if dlo <= hi and dhi >= lo:
                    found.append((kernel, sline, body.split("//")[0].strip()))
        m = st.match(line)
        if m:
            pending = (int(m.group(1)), int(m.group(2)), body.split("//")[0].strip())
    return found


def offload_arch(flags=None):
    """The one GPU target of the build, from its own flags (--offload-arch=...)."""
    archs = [f.split("=", 1)[1] for f in (flags or FLAGS) if f.startswith("--offload-arch=")]
    return archs[-1] if archs else "gfx950"


def device_disassembly(lib=None, arch=None):
    """The device code objects inside a built libsfmi.so (one per .hip source), disassembled: one string."""
    import tempfile

    lib = lib or LIB
    arch = arch or offload_arch()
    llvm = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = []
        k = blob.find(magic)
        while k >= 0:
            starts.append(k)
            k = blob.find(magic, k + 1)
        for n, a in enumerate(starts):
            b = starts[n + 1] if n + 1 < len(starts) else len(blob)
            part, co = os.path.join(td, "b%d.bin" % n), os.path.join(td, "b%d.co" % n)
            open(part, "wb").write(blob[a:b])
            subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--" + arch, "--input=" + part, "--output=" + co],
                                  stderr=subprocess.DEVNULL)
            out.append(subprocess.check_output([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=" + arch, co], text=True))
    return "\n".join(out)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bid = source_hash(extra_flags)
    # one compiler process per source, in parallel (the step kernel and the frame kernel take two minutes each), then the link
    import concurrent.futures
    import tempfile

    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags) + ['-DSFMI_BUILD_ID="%s"' % bid]
    with tempfile.TemporaryDirectory() as td:
        objs = [os.path.join(td, os.path.splitext(s)[0] + ".o") for s in SOURCES]

        def one(job):
            src, obj = job
            cmd = [hipcc] + cflags + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)

        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
            list(ex.map(one, zip(SOURCES, objs)))
        arch = offload_arch(cflags)
        cmd = [hipcc, "--offload-arch=" + arch, "-fPIC", "-shared", "--hip-link"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # the store-hazard scan needs llvm-objcopy, clang-offload-bundler and llvm-objdump: a library that could not be scanned is
    # removed like one that failed the scan -- never a .so without its .id (every later import would rebuild and fail again)
    try:
        bare = scan_wide_store_hazard(device_disassembly(LIB, arch))
    except (FileNotFoundError, subprocess.CalledProcessError) as e:
        os.remove(LIB)
        raise RuntimeError("libsfmi.so was built but its device code could not be scanned for the wide-store hazard (%s); "
                           "ROCM_LLVM_BIN names the directory of llvm-objcopy / clang-offload-bundler / llvm-objdump" % (e,))
    if bare:  # (see scan_wide_store_hazard: such a library plays wrong games in batches beyond 65 536 envs)
        os.remove(LIB)
        raise RuntimeError("libsfmi.so: %d wide buffer stores without their wait state, e.g. %s" % (len(bare), bare[0]))
    with open(LIB + ".id", "w") as f:
        f.write(bid + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
