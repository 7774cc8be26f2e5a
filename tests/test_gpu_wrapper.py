"""The HIP path against the reference's REAL `ssf_env.py`, EXECUTED (tests/golden/wrapper, made by make_wrapper_golden.py: the
reference's own file over its own CPython extension; see tests/test_wrapper_golden.py for what is stored and what is masked).

Bar: reward / done / info identical on every step of the 15 runs x 3 observation types; every observation column that is an
integer, a flag, a position, a velocity or a timer BIT-identical in float64; the three columns that come out of atan2 / sqrt
chains (aim, vdir, ndist and what normalized-features makes of them) within 1e-9 absolute (the device's libm is not glibc's, and
vdir's first angle is derived from the bearing instead of a second atan2: DESIGN 9) -- about 84 % of them are bit-identical and
the share is asserted (> 70 %); monitors, thresholds of them, are compared exactly.  Masked: kill_ready, the reference's
undefined getter (features / normalized-features column 12, monitors column 3).  The reset observation is compared in full with
SF_FLAG_REF_RESET_OBS (aim = vdir = ndist = 0, as the reference returns on fresh memory) -- for the first Game of every run
and for the auto-reset at done (a15: the new game's observation replaces the terminal one)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_wrapper_golden import EXTRAS, KILL_READY, OBS_TYPES, RUNS, WRAPPER, unmasked

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    from spacefortress_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsfmi.so not built: the GPU tests never fall back"
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


def _check(got, want, ot, what, stats):
    cols = unmasked(ot, want.shape[-1])
    soft = [c for c in EXTRAS[ot] if c in cols] if ot != "monitors" else []
    hard = [c for c in cols if c not in soft]
    assert np.array_equal(got[..., hard], want[..., hard]), (what, "exact columns", np.argwhere(got[..., hard] != want[..., hard])[:5].tolist())
    if soft:
        d = np.abs(got[..., soft] - want[..., soft])
        assert d.max() <= 1e-9, (what, "extras", float(d.max()))
        stats[0] += int((got[..., soft] == want[..., soft]).sum())
        stats[1] += d.size


@pytest.mark.parametrize("name", RUNS)
def test_hip_equals_the_executed_wrapper(sfa, name):
    z = np.load(os.path.join(WRAPPER, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    acts = np.load(os.path.join(GOLDEN, name + ".npz"))["actions"]
    done = z["done"].astype(bool)
    stats = [0, 0]
    for ot in OBS_TYPES:
        k = ot.replace("-", "_")
        want, want_reset = z["obs_" + k].copy(), z["reset_" + k]
        # the VecEnv view (a15): at done the new game's observation replaces the terminal one
        vec = want.copy()
        vec[done] = want_reset[1:]
        for ref_reset in (True, False):
            env = sfa.SFVecEnv(1, gametype=meta["gametype"], obs_type=ot, action_set=meta["action_set"], seed=meta["seed"],
                               spawn_skip=meta["spawn_skip"], obs_dtype=torch.float64, ref_reset_obs=ref_reset)
            # no reset(): the run starts from the first Game, which sf_create made
            a = torch.from_numpy(acts[:, None].astype(np.uint8)).to(env.device)
            obs, rew, dn, info = env.rollout(a)
            obs, rew, dn, info = obs.cpu().numpy()[:, 0], rew.cpu().numpy()[:, 0], dn.cpu().numpy()[:, 0], info.cpu().numpy()[:, 0]
            assert np.array_equal(rew, z["reward"]) and np.array_equal(dn, z["done"]) and np.array_equal(info, z["info"]), (name, ot)
            if ref_reset:
                _check(obs, vec, ot, (name, ot, "steps + auto-reset"), stats)
            else:  # the default: computeExtra(spawn) in a new game's observation -- only those columns differ from the reference
                _check(obs[~done], vec[~done], ot, (name, ot, "steps"), stats)
                if done.any():
                    keep = [c for c in unmasked(ot, vec.shape[1]) if c not in EXTRAS[ot]]
                    assert np.array_equal(obs[done][:, keep], vec[done][:, keep])
                    if ot == "features":
                        assert (obs[done][:, EXTRAS[ot]] != 0).all() and (vec[done][:, EXTRAS[ot]] == 0).all()
            env.close()
    assert stats[1] > 0 and stats[0] / stats[1] > 0.7, stats  # (measured: 84 % of the extras' values are bit-identical, all within 1e-9)


@pytest.mark.parametrize("name", ["youturn_random_ep", "autoturn_destroy", "youturn_hunter"])
def test_ssf_env_equals_the_executed_wrapper(sfa, name):
    """The single-env surface (SSF_Env: no auto-reset, like the reference's class): step returns the TERMINAL observation at
    done, reset() the new game's; np_random exists from the constructor on (ENV:53)."""
    z = np.load(os.path.join(WRAPPER, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    assert meta["spawn_skip"] == 0
    acts = np.load(os.path.join(GOLDEN, name + ".npz"))["actions"]
    stats = [0, 0]
    for ot in OBS_TYPES:
        k = ot.replace("-", "_")
        want, want_reset = z["obs_" + k], z["reset_" + k]
        env = sfa.SSF_Env(meta["gametype"], action_set=meta["action_set"], obs_type=ot, seed=meta["seed"], ref_reset_obs=True)
        assert env.np_random is not None
        n_reset = 0
        T = len(acts) if name != "youturn_hunter" else 600
        for t in range(T):
            o, r, d, i = env.step(int(acts[t]))
            assert (r, d, i) == (int(z["reward"][t]), bool(z["done"][t]), bool(z["info"][t])) and isinstance(i, bool), (name, ot, t)
            _check(np.asarray(o, np.float64), want[t], ot, (name, ot, t), stats)
            if d:
                n_reset += 1
                _check(np.asarray(env.reset(), np.float64), want_reset[n_reset], ot, (name, ot, "reset", n_reset), stats)
        assert n_reset == (len(want_reset) - 1 if T == len(acts) else 0)
        env.close()


def test_sf_reset_returns_the_references_first_observation(sfa):
    """youturn_seed12345_skip3 was recorded on the FOURTH Game of its libc stream: a batch created with spawn_skip = 2 holds the
    third, and its reset() -- sf_reset's observation -- is the run's first Game, whose observation the reference's reset()
    returned inside __init__ (fixture: reset_*[0]).  The run then replays from there."""
    name = "youturn_seed12345_skip3"
    z = np.load(os.path.join(WRAPPER, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    assert meta["spawn_skip"] == 3
    acts = np.load(os.path.join(GOLDEN, name + ".npz"))["actions"]
    stats = [0, 0]
    for ot in OBS_TYPES:
        k = ot.replace("-", "_")
        env = sfa.SFVecEnv(1, gametype=meta["gametype"], obs_type=ot, action_set=meta["action_set"], seed=meta["seed"],
                           spawn_skip=2, obs_dtype=torch.float64, ref_reset_obs=True)
        first = env.reset().cpu().numpy()[0]
        _check(first, z["reset_" + k][0], ot, (ot, "first observation"), stats)
        if ot == "features":
            assert (first[EXTRAS[ot]] == 0).all()
        obs, rew, dn, info = env.rollout(torch.from_numpy(acts[:, None].astype(np.uint8)).to(env.device))
        assert np.array_equal(rew.cpu().numpy()[:, 0], z["reward"])
        _check(obs.cpu().numpy()[:, 0], z["obs_" + k], ot, (ot, "steps"), stats)
        env.close()
