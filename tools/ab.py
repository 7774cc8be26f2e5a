#!/usr/bin/env python3
"""A/B timing of differently built libsfmi variants on ONE device, interleaved rounds
(guide rule 24: perf deltas come from interleaved rounds, not from separate invocations).

    python tools/ab.py build/abl/libsfmi_a.so build/abl/libsfmi_b.so [--envs 65536] [--rounds 5]

Each variant runs in its own subprocess per round (a process can only load one libsfmi); rounds
interleave the variants; prints median and min us/step per variant.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--gametype", default="youturn")
    ap.add_argument("--obs-type", default="features")
    ap.add_argument("--launch", default="auto", choices=("auto", "loop", "graph"),
                    help="bench.py --launch: one launch per step from the Python loop, or HIP graphs of 256 launches")
    a = ap.parse_args()
    res = {l: [] for l in a.libs}
    for r in range(a.rounds):
        for l in a.libs:
            env = dict(os.environ, SFMI_LIB_PATH=os.path.abspath(l))
            out = subprocess.check_output(
                [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "300",
                 "--no-cpu-baseline", "--envs", str(a.envs), "--gametype", a.gametype, "--obs-type", a.obs_type,
                 "--rollout-k", "0", "--image-envs", "0", "--kernel-timing-launches", "1", "--repeats", "1", "--numpy-api", "0",
                 "--no-configs", "--steady-seconds", "0.2", "--launch", a.launch],
                env=env, stderr=subprocess.DEVNULL, text=True)
            res[l].append(json.loads(out.strip().splitlines()[-1])["ms_per_step"] * 1e3)
    for l, v in res.items():
        v = sorted(v)
        print("%-40s median %.2f us  min %.2f us  (%s)" % (os.path.basename(l), v[len(v) // 2], v[0],
                                                          " ".join("%.2f" % x for x in v)))


if __name__ == "__main__":
    main()
