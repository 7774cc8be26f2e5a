"""Registers, scratch and LDS of every kernel in libsfmi.so (the code objects' metadata notes)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import build as B

llvm = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
lib = sys.argv[1] if len(sys.argv) > 1 else B.LIB
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, blob)]
    for n, a in enumerate(starts):
        b = starts[n + 1] if n + 1 < len(starts) else len(blob)
        part, co = os.path.join(td, "b%d.bin" % n), os.path.join(td, "b%d.co" % n)
        open(part, "wb").write(blob[a:b])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + part, "--output=" + co], stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", co], text=True)
        for blk in notes.split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            if len(sys.argv) > 2 and sys.argv[2] not in name:
                continue
            print("%-70s vgpr %4s agpr %4s sgpr %4s scratch %6s lds %6s spill_v %s" % (name[:70], g("vgpr_count"), g("agpr_count") if "agpr_count" in blk else blk.split()[1], g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count")))
