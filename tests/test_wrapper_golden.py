"""The wrapper rows (a13 `SSF_Env.step`, a14 `_get_features`, a15 reset on done) against the reference's REAL
`ssf_env.py`, EXECUTED: tests/golden/wrapper/*.npz hold what its step / reset returned over the reference's own CPython
extension for the 15 recorded runs x (features, normalized-features, monitors) -- 79 953 steps
(tests/golden/wrapper/make_wrapper_golden.py: the file loaded by path, import-only stand-ins for gym / pyglet / cv2).
Here: the oracle's restatement (oracle/sf_oracle.c:687-783) equals them, value for value in float64.  The HIP path is held
to the same files in tests/test_gpu_wrapper.py.

Masked, each for its reason (the fixtures' meta says the same):
  * kill_ready -- features / normalized-features column 12, monitors column 3: the reference feeds an int to
    Py_BuildValue("d") (SRC/pymodule.cpp:43-44), undefined behaviour; oracle and product define it as intended
    (vlner > 10 and the vulnerability timer below its limit: SURVEY 8a note 2);
  * nothing else.  The reset observation's aim / vdir / ndist (unwritten mExtra: zeros on fresh memory) are reproduced with
    `ref_reset_obs`; without it they are computeExtra(spawn) (SURVEY 8a note 4) and only those columns differ.
"""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

WRAPPER = os.path.join(GOLDEN, "wrapper")
RUNS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(WRAPPER, "*.npz")))
OBS_TYPES = ("features", "normalized-features", "monitors")
KILL_READY = {"features": 12, "normalized-features": 12, "monitors": 3}
EXTRAS = {"features": [6, 7, 8], "normalized-features": [6, 7, 8], "monitors": [4, 5, 6, 7, 8, 9]}


def unmasked(obs_type, dim):
    return [c for c in range(dim) if c != KILL_READY[obs_type]]


def test_the_fixture_set_is_complete():
    assert len(RUNS) == 15
    total = 0
    for name in RUNS:
        z = np.load(os.path.join(WRAPPER, name + ".npz"))
        meta = json.loads(str(z["meta"]))
        assert meta["executed"].endswith("spacefortress/gym/envs/ssf_env.py") and "import-only" in meta["stand_ins"]
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        assert np.array_equal(z["reward"], g["reward"]) and np.array_equal(z["done"], g["done"]) and np.array_equal(z["info"], g["info"])
        for ot in OBS_TYPES:
            k = ot.replace("-", "_")
            D = 10 if ot == "monitors" else meta["obs_dim"]
            assert z["obs_" + k].shape == (len(g["actions"]), D) and z["reset_" + k].shape == (1 + int(g["done"].sum()), D)
            assert z["obs_" + k].dtype == np.float64
        total += len(g["actions"])
    assert total == 26651


@pytest.mark.parametrize("name", RUNS)
def test_oracle_equals_the_executed_wrapper(oracle_mod, name):
    O = oracle_mod
    z = np.load(os.path.join(WRAPPER, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    acts = np.load(os.path.join(GOLDEN, name + ".npz"))["actions"]
    for ot in OBS_TYPES:
        k = ot.replace("-", "_")
        want, want_reset = z["obs_" + k], z["reset_" + k]
        cols = unmasked(ot, want.shape[1])
        for ref_reset in (True, False):
            e = O.OracleEnv(meta["gametype"], action_set=meta["action_set"], obs_type=ot, seed=meta["seed"], spawn_skip=meta["spawn_skip"])
            # (the oracle's constructor makes the first Game before the flag can be set: its extras' columns are compared on
            #  the later resets -- the two *_random_ep runs -- and, for every run, on the GPU, where the flag is a create flag)
            e.set_ref_reset_obs(ref_reset)
            rcols = cols if ref_reset else [c for c in cols if c not in EXTRAS[ot]]
            first, fcols = e.features(), [c for c in cols if c not in EXTRAS[ot]]
            assert np.array_equal(first[fcols], want_reset[0][fcols]), (name, ot, "first observation")
            assert (want_reset[:, EXTRAS["features"]] == 0).all() if ot == "features" else True  # fresh memory: zeros
            n_reset = 0
            for t, a in enumerate(acts):
                o, r, d, i = e.step(int(a))
                assert (r, d, i) == (int(z["reward"][t]), bool(z["done"][t]), bool(z["info"][t])), (name, ot, t)
                assert np.array_equal(o[cols], want[t][cols]), (name, ot, t, o, want[t])
                if d:
                    n_reset += 1
                    ro = e.reset()
                    assert np.array_equal(ro[rcols], want_reset[n_reset][rcols]), (name, ot, "reset", n_reset, ro, want_reset[n_reset])
                    if not ref_reset and ot != "monitors":
                        assert not np.array_equal(ro[EXTRAS[ot]], want_reset[n_reset][EXTRAS[ot]])  # the documented difference
            assert n_reset == len(want_reset) - 1


def test_kill_ready_is_the_only_undefined_column():
    """What the reference's undefined getter produced here, for the record: whenever the intended predicate's first half
    (vlner > 10) is false the wrapper's `and` short-circuits to 0 like ours; where it is true the reference's value depends on
    a stale register -- the fixtures hold whatever came out."""
    seen = set()
    for name in RUNS:
        z = np.load(os.path.join(WRAPPER, name + ".npz"))
        f = z["obs_features"]
        low = f[:, 11] <= 10
        assert (f[low, 12] == 0).all()
        seen |= set(np.unique(f[~low, 12]).tolist())
    assert seen <= {0.0, 1.0}
