"""The per-tick event bitmask (sf_set_event_output, SF_EV_*) against the event strings of the REAL
reference engine recorded in tests/golden/telemetry/dumps.npz: for every tick of three recorded runs the
set of game events must be the same; key bits must be the key-state changes of the tick."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("name", ["autoturn_destroy", "youturn_deaths", "youturn_rapid_fire"])
def test_event_masks_match_reference_event_strings(name):
    from spacefortress_amd import SFVecEnv, _lib
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    dumps = np.load(os.path.join(GOLDEN, "telemetry", "dumps.npz"))[name]
    meta = json.loads(str(z["meta"]))
    N = 3
    env = SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"], spawn_skip=meta["spawn_skip"])
    ev = env.enable_events()
    names = _lib.EVENT_NAMES
    prev_keys = 0
    seen = set()
    for t, a in enumerate(z["actions"]):
        env.step_tensors(torch.full((N,), int(a), dtype=torch.uint8, device=env.device))
        m = ev.cpu().numpy().astype(np.uint32)
        assert (m == m[0]).all()
        got = {names[b] for b in names if int(m[0]) & b}
        ref = set(re.findall(r'"([a-z\\-]+)"', dumps[t + 1].decode()))
        keys = int(z["keys"][t])
        want_keys = set()
        for bit, nm in ((1, "fire"), (2, "thrust"), (4, "left"), (8, "right")):
            if (keys & bit) and not (prev_keys & bit):
                want_keys.add("press-" + nm)
            if not (keys & bit) and (prev_keys & bit):
                want_keys.add("release-" + nm)
        prev_keys = keys
        game_ref = {e for e in ref if not e.startswith(("press-", "release-"))}
        game_got = {e for e in got if not e.startswith(("press-", "release-")) and e not in ("missile-left", "game-over")}
        assert game_got == game_ref, (t, game_got, game_ref)
        assert {e for e in got if e.startswith(("press-", "release-"))} == want_keys, t
        assert want_keys <= ref  # the reference logged those calls too
        seen |= game_got
    if name == "autoturn_destroy":
        assert {"missile-fired", "hit-fortress", "vlner-increased", "fortress-destroyed"} <= seen
    if name == "youturn_deaths":
        assert {"explode-bighex", "ship-respawn", "fortress-fired"} <= seen
    # the fused launch writes one row per tick
    K = 16
    acts = torch.from_numpy(np.repeat(z["actions"][:K, None], N, 1).astype(np.uint8)).to(env.device)
    e2 = SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"], spawn_skip=meta["spawn_skip"])
    e2.enable_events()
    e2.rollout(acts)
    e3 = SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"], spawn_skip=meta["spawn_skip"])
    ev3 = e3.enable_events()
    for k in range(K):
        e3.step_tensors(acts[k])
        assert torch.equal(e2.rollout_events[k], ev3)
    for e in (env, e2, e3):
        e.close()


@pytest.mark.gpu
def test_sampled_rollout_with_events_enabled():
    """enable_events() then rollout_sampled(K): the fused launch stores K rows of event masks, so it must get a [K, N]
    buffer of its own (`rollout_events`) instead of overrunning the one-row `events` tensor; the rows equal what K
    step_sampled() calls write, and a canary tensor allocated right behind `events` stays untouched."""
    from spacefortress_amd import SFVecEnv

    N, K = 256, 12
    e1 = SFVecEnv(N, gametype="youturn", seed=3, spawn_stride=1)
    e2 = SFVecEnv(N, gametype="youturn", seed=3, spawn_stride=1)
    for e in (e1, e2):
        e.reset()
        e.seed_actions(99)
    ev1 = e1.enable_events()
    canary = torch.full((4 * K * N,), 0x5A5A5A5A, dtype=torch.int32, device=e1.device)
    ev2 = e2.enable_events()
    _, rew, done, info, acts = e1.rollout_sampled(K, want_obs=False)
    torch.cuda.synchronize()
    assert e1.rollout_events is not None and tuple(e1.rollout_events.shape) == (K, N)
    assert bool((canary == 0x5A5A5A5A).all())
    for k in range(K):
        a = torch.empty(N, dtype=torch.uint8, device=e2.device)
        _, r2, d2, i2 = e2.step_sampled(actions_out=a)
        assert torch.equal(a, acts[k]) and torch.equal(r2, rew[k]) and torch.equal(d2, done[k]) and torch.equal(i2, info[k])
        assert torch.equal(ev2, e1.rollout_events[k]), k
    # the one-row buffer is back in place for the next single step
    _, r1, _, _ = e1.step_sampled()
    _, r2, _, _ = e2.step_sampled()
    assert torch.equal(r1, r2) and torch.equal(ev1, ev2)
    e1.close()
    e2.close()
