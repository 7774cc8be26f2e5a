#!/usr/bin/env python3
"""SubprocVecEnv-shaped CPU baseline (TEST/BENCH INFRASTRUCTURE, never the product path).

The reference steps its envs through gym_vecenv.SubprocVecEnv (rl/train.py:30-32,80): one OS
process per env, a multiprocessing.Pipe per worker, every step pickles the action to the
worker and (obs, reward, done, info) back, the worker resets on done.  gym_vecenv is not in
this image; this is the same shape with the oracle's SSF_Env restatement (oracle/sf_oracle.c,
called through ctypes -- the reference's wrapper spends more time per step in Python than
this does) in each worker.  Prints one JSON line {"steps", "seconds", "procs"}.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(conn, gametype):
    from oracle import oracle as O

    env = O.OracleEnv(gametype, obs_type="features")
    while True:
        cmd, data = conn.recv()
        if cmd == "step":
            obs, r, d, info = env.step(data)
            if d:
                obs = env.reset()
            conn.send((obs, r, d, info))
        elif cmd == "reset":
            conn.send(env.reset())
        else:
            conn.close()
            return


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--gametype", default="youturn")
    ap.add_argument("--seconds", type=float, default=10.0)
    a = ap.parse_args()
    import numpy as np

    ctx = mp.get_context("fork")
    pipes, procs = [], []
    for _ in range(a.procs):
        parent, child = ctx.Pipe()
        p = ctx.Process(target=worker, args=(child, a.gametype), daemon=True)
        p.start()
        child.close()
        pipes.append(parent)
        procs.append(p)
    for c in pipes:
        c.send(("reset", None))
    for c in pipes:
        c.recv()
    rng = np.random.RandomState(0)
    n_act = 5 if a.gametype in ("youturn", "test-youturn") else 3
    steps = 0
    t0 = time.perf_counter()
    while True:
        acts = rng.randint(0, n_act, a.procs)
        for c, act in zip(pipes, acts):
            c.send(("step", int(act)))
        res = [c.recv() for c in pipes]
        np.stack([r[0] for r in res])
        steps += a.procs
        if steps % (a.procs * 256) == 0 and time.perf_counter() - t0 >= a.seconds:
            break
    dt = time.perf_counter() - t0
    for c in pipes:
        c.send(("close", None))
    for p in procs:
        p.join(2)
    print(json.dumps({"steps": steps, "seconds": dt, "procs": a.procs}))


if __name__ == "__main__":
    main()
