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
    ap.add_argument("--rollout-steps", type=int, default=0,
                    help="BASELINE.json configs[0]: ONE env, a rollout of exactly this many steps with actions from "
                         "numpy RandomState(0).randint(0, n_actions) through the engine's press / release / step_one_tick "
                         "entry points one call at a time (what SSF_Env.step does, ENV:208-253), timed")
    a = ap.parse_args()
    if a.rollout_steps:
        import numpy as np
        ref = a.kind == "reference" and O.have_ref()
        env = O.OracleEnv(a.gametype)  # the action table + wrapper logic; the engine under it is the restatement ...
        g = O.RefGame(a.gametype) if ref else env  # ... or the real reference engine
        keys = env.action_keys()
        acts = np.random.RandomState(0).randint(0, len(keys), a.rollout_steps)
        best = None
        for rep in range(5):
            g.new_game()
            ret = 0
            t0 = time.perf_counter()
            for act in acts:
                g.apply_keys(keys[act], env.youturn)
                ret += g.step_one_tick(34)
                if g.is_game_over():
                    g.new_game()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"steps": int(a.rollout_steps), "seconds": best, "kind": "reference" if ref else "port",
                          "engine_return": int(ret)}))
        return
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
