#!/usr/bin/env python3
"""Batches beyond one wave per SIMD against the regime the oracle soaks cover.

Up to 65 536 envs every wave of sf_step_kernel is alone on its SIMD, and that is where every lock-step run against the CPU
oracle lives (tools/soak.py: the oracle steps 2e6 envs/s).  Beyond, several waves share a SIMD and the hardware's timing is
another: the wide-store hazard of round 4 (sf_buf_st128) only showed there.  This run plays a BIG batch (default 262 144
envs) and the same envs as batches of 65 536 -- same spawns, same actions, chunks of fused rollouts and of single steps
alternately, hunter play with 10 % random actions, past the episode's end -- and compares every reward / done / info /
observation of every step and every state field at every chunk end, bit for bit, on the device.

    python tools/big_batch_soak.py [--envs 262144] [--steps 6000] [--gametype youturn] [--streams]

--streams: additionally the small batches step CONCURRENTLY on separate HIP streams (their waves share SIMDs with each
other's) and must still play the same games.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from spacefortress_amd import SFVecEnv  # noqa: E402
from sfscript import HUNTER_PATTERN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=262144)
    ap.add_argument("--small", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--chunk", type=int, default=200)
    ap.add_argument("--gametype", default="youturn")
    ap.add_argument("--obs-type", default="features")
    ap.add_argument("--streams", action="store_true")
    a = ap.parse_args()
    N, n, K = a.envs, a.small, a.chunk
    assert N % n == 0
    nb = N // n
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    big = SFVecEnv(N, gametype=a.gametype, obs_type=a.obs_type, spawn_stride=1)
    small = [SFVecEnv(n, gametype=a.gametype, obs_type=a.obs_type, spawn_stride=1, spawn_skip=k * n) for k in range(nb)]
    streams = [torch.cuda.Stream() for _ in range(nb)] if a.streams else None
    pat = torch.from_numpy(HUNTER_PATTERN).to(dev)
    phase = torch.randint(0, len(HUNTER_PATTERN), (N,), device=dev, generator=g)
    t0 = time.time()
    steps = kills = episodes = 0
    for c in range(0, a.steps, K):
        k = min(K, a.steps - c)
        rnd = torch.randint(0, big.n_actions, (k, N), device=dev, generator=g, dtype=torch.int64).to(torch.uint8)
        tt = (torch.arange(c, c + k, device=dev)[:, None] + phase[None, :]) % len(HUNTER_PATTERN)
        acts = torch.where(torch.rand((k, N), device=dev, generator=g) < 0.1, rnd, pat[tt]).contiguous()
        fused = (c // K) % 2 == 0

        def play(env, act):
            if fused:
                return env.rollout(act)
            outs = [tuple(x.clone() for x in env.step_tensors(act[j])) for j in range(act.shape[0])]
            return tuple(torch.stack([o[j] for o in outs]) for j in range(4))

        ob, rb, db, ib = play(big, acts)
        parts = [acts[:, j * n:(j + 1) * n].contiguous() for j in range(nb)]
        torch.cuda.synchronize()
        if streams:
            res = [None] * nb
            for j in range(nb):
                with torch.cuda.stream(streams[j]):
                    res[j] = play(small[j], parts[j])
            torch.cuda.synchronize()
        else:
            res = [play(small[j], parts[j]) for j in range(nb)]
        for j in range(nb):
            os_, rs, ds, is_ = res[j]
            sl = slice(j * n, (j + 1) * n)
            for name, x, y in (("reward", rb[:, sl], rs), ("done", db[:, sl], ds), ("info", ib[:, sl], is_)):
                if not torch.equal(x, y):
                    w = torch.nonzero(x != y)[:6].tolist()
                    raise SystemExit("MISMATCH %s chunk at step %d batch %d: (tick, lane) %s" % (name, c, j, w))
            if not torch.equal(ob[:, sl].view(torch.int32), os_.view(torch.int32)):  # bit for bit
                w = torch.nonzero(ob[:, sl].view(torch.int32) != os_.view(torch.int32))[:6].tolist()
                raise SystemExit("MISMATCH obs chunk at step %d batch %d: (tick, lane, feature) %s" % (c, j, w))
        sb = big.state_dict()
        for j in range(nb):
            sd = small[j].state_dict()
            for key in sd:
                x, y = np.asarray(sb[key])[..., j * n:(j + 1) * n], np.asarray(sd[key])
                if x.tobytes() != y.tobytes():
                    raise SystemExit("MISMATCH state %s after step %d batch %d: %s" % (key, c + k, j, np.argwhere(x != y)[:6].tolist()))
        steps += k
        kills += int(ib.sum())
        episodes += int(db.sum())
        del ob, rb, db, ib, res
        print("step %6d  %s  kills %d  episodes %d  %.0f s" % (steps, "fused " if fused else "single", kills, episodes, time.time() - t0), flush=True)
    print("OK: %d envs x %d steps = %.2e env-steps in one batch == %d batches of %d%s (%s, %s): every output, every state field"
          % (N, steps, N * steps, nb, n, ", concurrent on %d streams" % nb if streams else "", a.gametype, a.obs_type))


if __name__ == "__main__":
    main()
