#!/usr/bin/env python3
"""Wide buffer stores without their wait state (spacefortress_amd/build.py: scan_wide_store_hazard) in a built libsfmi.so
or in assembly files (hipcc --cuda-device-only -S):

    python tools/store_hazard_scan.py                        # the in-tree library
    python tools/store_hazard_scan.py build/abl/libsfmi_x.so k.s ...
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import build as B  # noqa: E402


def main():
    n = 0
    for path in sys.argv[1:] or [B.LIB]:
        text = B.device_disassembly(path) if path.endswith(".so") else open(path).read()
        found = B.scan_wide_store_hazard(text)
        for kernel, st, nxt in found:
            print("%s: %s\n    %s\n    %s" % (path, kernel, st, nxt))
        print("%s: %d hazardous wide buffer stores, %d wide buffer stores in all" % (path, len(found), text.count("buffer_store_dwordx4") + text.count("buffer_store_dwordx3")))
        n += len(found)
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main())
