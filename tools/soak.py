"""Long lock-step parity run: N lanes x T steps (several episodes) on the GPU (chunks of fused rollouts and,
alternately, single steps) against the CPU oracle; every reward / done / info, every observation (float32
tolerance), the full state at every chunk end.  python tools/soak.py [gametype] [lanes] [steps] [random|hunter|charger] [obs_type] [f64]
`hunter`: an open-loop firing pattern per lane (a shot every 8 ticks = 272 ms > the 250 ms vulnerability window
until the fortress is kill-ready, then a double shot), random phase per lane, 10 % of the actions random: thousands
of fortress kills, resets and misses instead of the handful random play produces.
(The oracle steps 1.7e6 envs/s on one core: 65 536 lanes x 3 000 steps take two minutes.  Bigger batches:
tools/big_batch_soak.py, device against device.)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from spacefortress_amd import SFVecEnv
from oracle import oracle as O
from sfcompare import compare_state

gametype = sys.argv[1] if len(sys.argv) > 1 else "youturn"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16500
policy = sys.argv[4] if len(sys.argv) > 4 else "random"
obs_type = sys.argv[5] if len(sys.argv) > 5 else "features"  # features | normalized-features | monitors
f64 = len(sys.argv) > 6 and sys.argv[6] == "f64"             # float64 observations, compared to 1e-9
K = 250
env = SFVecEnv(N, gametype=gametype, obs_type=obs_type, spawn_stride=3, spawn_skip=1,
               obs_dtype=torch.float64 if f64 else torch.float32)
orc = O.OracleVecEnv(gametype, N, obs_type=obs_type, spawn_stride=3, spawn_skip=1)
RT, AT = (1e-9, 1e-9) if f64 else (1e-5, 4e-5)
rng = np.random.default_rng(99)
phase = rng.integers(0, 96, N)
o0 = env.reset().cpu().numpy(); oo0 = orc.reset()
assert np.allclose(o0, oo0, rtol=RT, atol=AT)
t0 = time.time(); done_total = 0; kills = 0
for c in range(0, T, K):
    k = min(K, T - c)
    acts = rng.integers(0, env.n_actions, (k, N)).astype(np.uint8)
    if policy == "charger":  # THRUST (action 2 in both action sets) half of the time: in autoturn games the ship flies at
        # the fortress along an exact-degree ray, the regime where ceil(bearing) follows atan2's last bit
        acts = np.where(rng.random((k, N)) < 0.5, np.uint8(2), acts).astype(np.uint8)
    if policy == "hunter":  # FIRE is action 1 in both action sets (ENV:211-229)
        pat = np.array(([1] + [0] * 7) * 11 + [1, 0, 1, 0] + [0] * 4, np.uint8)
        tt = (np.arange(c, c + k)[:, None] + phase[None, :]) % len(pat)
        acts = np.where(rng.random((k, N)) < 0.1, acts, pat[tt]).astype(np.uint8)
    a = torch.from_numpy(acts).to(env.device)
    if (c // K) % 2 == 0:
        obs, rew, done, info = env.rollout(a)
    else:
        outs = [tuple(x.clone() for x in env.step_tensors(a[j])) for j in range(k)]
        obs, rew, done, info = (torch.stack([o[j] for o in outs]) for j in range(4))
    obs, rew, done, info = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool), info.cpu().numpy().astype(bool)
    for j in range(k):
        oo, orw, od, oi = orc.step(acts[j].astype(np.int32))
        assert np.array_equal(rew[j], orw), (c + j, np.flatnonzero(rew[j] != orw)[:5])
        assert np.array_equal(done[j], od) and np.array_equal(info[j], oi), c + j
        ok = np.isclose(obs[j], oo, rtol=RT, atol=AT)
        if not ok.all():
            bad = np.argwhere(~ok)
            for lane, feat in bad[:8]:
                print("OBS MISMATCH step %d lane %d feature %d: device %r oracle %r   (device row %s)" % (
                    c + j, lane, feat, obs[j][lane, feat], oo[lane, feat], np.array2string(obs[j][lane], precision=6)), flush=True)
            raise AssertionError(c + j)
        done_total += int(od.sum()); kills += int(oi.sum())
    bad = compare_state(env.state_dict(), orc.snapshots())
    assert not bad, (c, bad)
    env.check_state()  # (the sticky error counters: a packed field that wrapped, a split launch's poll that gave up)
    if (c // K) % 10 == 9 or N * K >= 4000000:  # (big batches: every chunk takes seconds, say so)
        print("step %6d ok  (episodes finished %d, kills %d, %.0f s)" % (c + k, done_total, kills, time.time() - t0), flush=True)
print("SOAK OK: %s (%s), %d lanes x %d steps = %.1fM env-steps, %d episodes finished, %d fortress kills" % (
    gametype, policy, N, T, N * T / 1e6, done_total, kills))
