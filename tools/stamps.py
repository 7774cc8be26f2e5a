#!/usr/bin/env python3
"""Where does a wave of sf_step_kernel spend its cycles?  DIAGNOSTIC ONLY.

Builds libsfmi with -DSF_STAMPS into build/diag/ (shader-clock stamps at phase boundaries, with
forced waits so each phase owns its memory latency), steps a batch, and prints the median
per-phase cycles over all waves plus the first-wave-start -> last-wave-end span in real time.
Read the SHARES, not the length: the forced waits forbid overlaps the real kernel has.

    python tools/stamps.py [--envs 65536] [--gametype youturn]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "build", "diag", "libsfmi_stamps.so")


def build(lite=False):
    global DIAG
    if lite:
        DIAG = DIAG.replace("_stamps.so", "_stamps_lite.so")
    os.makedirs(os.path.dirname(DIAG), exist_ok=True)
    csrc = os.path.join(ROOT, "spacefortress_amd", "csrc")
    B = __import__("spacefortress_amd.build", fromlist=["SOURCES"])
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DSF_STAMPS"] + (["-DSF_STAMPS_LITE"] if lite else []) + [
        os.path.join(csrc, f) for f in B.SOURCES] + ["-o", DIAG]
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--gametype", default="youturn")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--lite", action="store_true", help="only real-time start/end per wave (no forced waits): "
                    "first-wave-start -> last-wave-end span of an otherwise unperturbed kernel")
    a = ap.parse_args()
    global DIAG
    if a.lite:
        DIAG = DIAG.replace("_stamps.so", "_stamps_lite.so")
    if os.environ.get("SF_STAMPS_LIB"):  # a -DSF_STAMPS build made elsewhere (tools/variant.py NAME -DSF_STAMPS ...)
        DIAG = os.path.abspath(os.environ["SF_STAMPS_LIB"])
    elif not os.path.exists(DIAG) or a.build_only:
        DIAG = DIAG.replace("_stamps_lite.so", "_stamps.so")
        build(a.lite)
    if a.build_only:
        return
    os.environ["SFMI_LIB_PATH"] = DIAG
    import numpy as np
    import torch
    from spacefortress_amd import SFVecEnv, _lib

    env = SFVecEnv(a.envs, gametype=a.gametype, spawn_stride=1, reuse_buffers=True)
    acts = torch.randint(0, env.n_actions, (64, a.envs), device=env.device, dtype=torch.uint8)
    for t in range(a.steps):
        env.step_tensors(acts[t % 64])
    torch.cuda.synchronize()
    L = _lib.lib()
    n_waves = (a.envs + 255) // 256 * 4
    buf = np.zeros((n_waves, 16), np.uint64)
    L.sf_debug_read.argtypes = [C.c_void_p, C.c_void_p]
    assert L.sf_debug_read(env._h, buf.ctypes.data_as(C.c_void_p)) == 0
    if a.lite:
        rt = buf[:, 12:14].astype(np.int64)
        life = (rt[:, 1] - rt[:, 0]) / 100.0
        print("lite: %d waves; span first start -> last end %.2f us; wave life median %.2f p90 %.2f max %.2f us; "
              "start spread %.2f us; end spread (last - median end) %.2f us" % (
                  n_waves, (rt[:, 1].max() - rt[:, 0].min()) / 100.0, np.median(life), np.percentile(life, 90), life.max(),
                  (rt[:, 0].max() - rt[:, 0].min()) / 100.0, (rt[:, 1].max() - np.median(rt[:, 1])) / 100.0))
        return
    s = buf[:, :10].astype(np.int64)
    names = ["issue loads (round trip 1)", "wait round trip 1", "prefetch issue + LDS stage + barrier",
             "keys/ship/atan2 x2/fortress", "wait projectile prefetch", "shells + missiles",
             "timers/reward/stats atomics/stores issue", "obs: extras + LDS transpose + flush", "drain stores"]
    d = np.diff(s, axis=1)
    tot = s[:, 9] - s[:, 0]
    print("waves %d; per-wave cycles (median / p90):" % n_waves)
    for k, nm in enumerate(names):
        print("  %-45s %7.0f %7.0f  (%4.1f%%)" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 90),
                                                100 * np.median(d[:, k]) / np.median(tot)))
    if buf[:, 10].any():  # sub-stamps inside "timers ... stores issue": 6 -> 10 -> 11 -> 14 -> 15 -> 7
        seq = buf[:, [6, 10, 11, 14, 15, 7]].astype(np.int64)
        print("  inside 'timers ... stores issue': timers+reward %d | counter atomics %d | lane stores %d | "
              "reward/done/info stores %d | trainer epilogue %d" % tuple(np.median(np.diff(seq, axis=1), axis=0)))
    print("  %-45s %7.0f %7.0f   p99 %.0f  max %.0f" % ("wave total", np.median(tot), np.percentile(tot, 90),
                                                   np.percentile(tot, 99), tot.max()))
    slow = np.argsort(tot)[-8:]
    print("  slowest waves, cycles per phase:")
    for w in slow:
        print("   wave %4d total %6d :" % (w, tot[w]), " ".join("%5d" % x for x in d[w]))
    rt = buf[:, 12:14].astype(np.int64)  # 100 MHz
    print("real time: first wave start -> last wave end %.2f us; median wave life %.2f us; start spread %.2f us" % (
        (rt[:, 1].max() - rt[:, 0].min()) / 100.0, np.median(rt[:, 1] - rt[:, 0]) / 100.0,
        (rt[:, 0].max() - rt[:, 0].min()) / 100.0))
    print("shader clock estimate: %.2f GHz" % (np.median(tot) / np.median(rt[:, 1] - rt[:, 0]) / 10.0))


if __name__ == "__main__":
    main()
