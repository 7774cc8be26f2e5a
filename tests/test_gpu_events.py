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
