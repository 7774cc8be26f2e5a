#!/usr/bin/env python3
"""Wrapper-level fixtures from the reference's REAL `ssf_env.py`, executed.

Run in the build container only (needs /root/reference and /opt/conda's cairo for `make -C oracle refpy`):

    python tests/golden/wrapper/make_wrapper_golden.py          ->  tests/golden/wrapper/<name>.npz, one per recorded run

What runs: the reference's own file ENV = python/spacefortress.gym/spacefortress/gym/envs/ssf_env.py, loaded BY PATH from where
it lies, its `SSF_Env.__init__` / `reset` / `step` / `_get_features` (ENV:50-93,95-157,163-178,208-253) over the reference's own
CPython extension (`spacefortress.core` = oracle/_ref/_spacefortress<ext>.so = SRC/pymodule.cpp + the engine, compiled where they
lie by oracle/Makefile).  The file imports `gym`, `pyglet` and `cv2` at module level; none is in this image and there is no
network, so three IMPORT-ONLY stand-in modules satisfy those lines: `gym.Env` (an empty base class), `gym.spaces.Box / Discrete`
(records of their arguments), `gym.utils.seeding.np_random` (numpy's RandomState), empty `pyglet` and `cv2`.  Nothing on the path
recorded here calls into them: `step`, `reset` and `_get_features` of the three symbolic observation types touch numpy, `copy`
and the extension only (the 'image' type would call cv2.cvtColor and is not recorded; its pixels are tests/golden/frames').
`spacefortress.core` is provided as a module that re-exports the extension (the reference's own core/__init__.py does
`from _spacefortress import *`, a Python-2 implicit relative import).

For every recorded run of tests/golden/*.npz (15) and every obs_type in (features, normalized-features, monitors): a fresh
SSF_Env(gametype, action_set=..., obs_type=...) under the run's libc stream (initstate(seed), `spawn_skip` Games first: as
make_getters_golden.py), the run's actions replayed through env.step, `env.reset()` when done (what gym_vecenv's worker does,
rl/train.py:80).  Stored per run:  obs_<type> f64[T][D] (what step returned: the terminal observation included),
reset_<type> f64[1 + resets][D] (after __init__, then what each env.reset() returned), reward i32[T], done u8[T], info u8[T],
and meta (json).  The script asserts that reward / done / info equal the run's own record (tests/golden/<name>.npz: the engine
driven through oracle/ref_driver.cpp with the wrapper's arithmetic restated) -- executed and restated agree.

Two settings of the C library make a run the recorded one (as in make_getters_golden.py; neither touches reference code):
initstate(seed) -- every env of the reference is a process of its own with its own rand() stream --, and MALLOC_PERTURB_=255 so
that `new Game` reads zeros where Game::Game leaves members unwritten (SRC/game.cpp:78), as in a fresh process.
KNOWN UNDEFINED in the reference, recorded as they came out HERE and masked by every consumer:
  * `kill_ready` (features / normalized-features column 12, monitors column 3): vulnerability_timer / vulnerability_time hand an
    int to Py_BuildValue("d") (SRC/pymodule.cpp:43-44);
  * aim, vdir, ndist of a RESET observation (columns 6, 7, 8; monitors 4..9): mExtra is unwritten until the first tick.
"""
import ctypes
import glob
import importlib.util
import json
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
ROOT = os.path.dirname(os.path.dirname(GOLDEN))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
ENV_PY = "/root/reference/python/spacefortress.gym/spacefortress/gym/envs/ssf_env.py"
OBS_TYPES = ("features", "normalized-features", "monitors")


def stand_ins(sf):
    """The import-only modules ssf_env.py's first lines ask for (see the docstring), and spacefortress.core."""
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class Box(object):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete(object):
        def __init__(self, n):
            self.n = n

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Discrete = Box, Discrete
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    utils.seeding = seeding
    gym.spaces, gym.utils = spaces, utils
    pkg = types.ModuleType("spacefortress")
    pkg.__path__ = []
    core = types.ModuleType("spacefortress.core")
    for k in dir(sf):
        if not k.startswith("__"):
            setattr(core, k, getattr(sf, k))
    pkg.core = core
    mods = {"gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding,
            "pyglet": types.ModuleType("pyglet"), "cv2": types.ModuleType("cv2"), "spacefortress": pkg, "spacefortress.core": core}
    for k in mods:
        assert k not in sys.modules, k + " is importable here: the stand-in would shadow it"
    sys.modules.update(mods)


def main():
    if os.environ.get("MALLOC_PERTURB_") != "255":
        env = dict(os.environ, MALLOC_PERTURB_="255")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refpy"])
    sys.path.insert(0, REFDIR)
    import _spacefortress as sf
    stand_ins(sf)
    spec = importlib.util.spec_from_file_location("ref_ssf_env", ENV_PY)
    E = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(E)  # the reference's file, executed where it lies
    libc = ctypes.CDLL(None)
    libc.initstate.restype = ctypes.c_void_p
    libc.initstate.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_size_t]
    keep = []
    total = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        name = os.path.basename(path)[:-4]
        if name == "tables":
            continue
        z = np.load(path)
        meta = json.loads(str(z["meta"]))
        acts = z["actions"]
        out = {}
        for ot in OBS_TYPES:
            rng = ctypes.create_string_buffer(128)
            keep.append(rng)  # (the C library writes into the state array it leaves when given a new one: never freed here)
            libc.initstate(meta["seed"], rng, 128)
            for _ in range(meta["spawn_skip"]):
                sf.Game(meta["gametype"], width=90, height=92, viewport=(130, 80, 450, 460), lw=3, grayscale=True)
            env = E.SSF_Env(gametype=meta["gametype"], action_set=meta["action_set"], obs_type=ot)  # __init__ -> reset() -> the run's first Game
            assert env.tickdur == 34 and env.max_ticks == 5294.0 and env.action_space.n == len(env.action_combinations)
            resets = [np.asarray(env._get_features(), np.float64)]  # (what that reset() returned: getters only)
            obs, rew, done, info = [], [], [], []
            for a in acts:
                o, r, d, i = env.step(int(a))
                obs.append(np.asarray(o, np.float64))
                rew.append(int(r))
                done.append(bool(d))
                info.append(bool(i))
                assert isinstance(i, (bool, np.bool_)), type(i)  # info is a bare bool (ENV:233,253)
                if d:
                    resets.append(np.asarray(env.reset(), np.float64))
            assert np.array_equal(rew, z["reward"]) and np.array_equal(done, z["done"].astype(bool)) and np.array_equal(info, z["info"].astype(bool)), \
                (name, ot, "the executed wrapper and the recorded run disagree")
            assert sum(env.actions_taken.values()) == len(acts)
            key = ot.replace("-", "_")
            out["obs_" + key], out["reset_" + key] = np.stack(obs), np.stack(resets)
            if "reward" not in out:
                out["reward"], out["done"], out["info"] = np.array(rew, np.int32), np.array(done, np.uint8), np.array(info, np.uint8)
            else:
                assert np.array_equal(out["reward"], rew) and np.array_equal(out["done"], done) and np.array_equal(out["info"], info)
        D = out["obs_features"].shape[1]
        out["meta"] = json.dumps(dict(meta, obs_types=list(OBS_TYPES), obs_dim=D, executed=ENV_PY,
                                      core="oracle/_ref/_spacefortress: SRC/pymodule.cpp + engine, compiled where they lie",
                                      stand_ins="import-only: gym (Env, spaces.Box, spaces.Discrete, utils.seeding.np_random), pyglet, cv2; "
                                                "spacefortress.core re-exports the extension",
                                      libc="initstate(seed) + spawn_skip Games first; MALLOC_PERTURB_=255",
                                      undefined="kill_ready (features col 12, monitors col 3): UB getter; reset obs aim/vdir/ndist "
                                                "(features cols 6-8, monitors cols 4-9): unwritten mExtra, zeros on fresh memory"))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        total += len(acts)
        print("%-28s %-13s T=%5d D=%2d resets=%d sum_r=%5d kills=%d  %5.0f KB" % (
            name, meta["gametype"], len(acts), D, len(out["reset_features"]) - 1, int(out["reward"].sum()), int(out["info"].sum()),
            os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))
    print("%d steps x %d observation types executed through %s" % (total, len(OBS_TYPES), ENV_PY))


if __name__ == "__main__":
    main()
