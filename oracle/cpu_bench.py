#!/usr/bin/env python3
"""CPU baseline worker (TEST/BENCH INFRASTRUCTURE, never the product path).

One process = one environment stepping with uniform random actions for about
--seconds of wall time; prints one JSON line {"steps", "seconds", "kind"}.
bench.py starts one of these per host core, which is the reference's own
parallel design (one process per env, rl/train.py:30-32).

kind "reference": the real reference engine (oracle/_ref/libsfref.so, bare
                  pressKey/releaseKey + stepOneTick loop of ref_driver.cpp);
kind "port":      the C restatement (oracle/libsforacle.so), same loop.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="reference")
    ap.add_argument("--gametype", default="youturn")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    if a.kind == "reference" and O.have_ref():
        g = O.RefGame(a.gametype)
        kind = "reference"
    else:
        g = O.OracleEnv(a.gametype)
        kind = "port"
    chunk = 100000
    g.rollout(chunk, a.seed)  # warm
    steps = 0
    t0 = time.perf_counter()
    while True:
        g.rollout(chunk, a.seed + 1 + steps // chunk)
        steps += chunk
        dt = time.perf_counter() - t0
        if dt >= a.seconds:
            break
    print(json.dumps({"steps": steps, "seconds": dt, "kind": kind}))


if __name__ == "__main__":
    main()
