"""Generates tests/golden/telemetry/dumps.npz: Game::dumpState() strings (SRC/game.cpp:519-576) of the REAL
reference engine (oracle/_ref) at every tick of three recorded runs whose action sequences are the
existing goldens'.  Build container only.

    python tests/golden/telemetry/make_dumps_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

out = {}
for name in ("autoturn_destroy", "youturn_deaths", "youturn_rapid_fire"):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    meta = json.loads(str(z["meta"]))
    g = O.RefGame(meta["gametype"], seed=meta["seed"], spawn_skip=meta["spawn_skip"])
    dumps, eng = [g.dump()], []
    for k in z["keys"]:
        g.apply_keys(int(k), g.youturn)
        eng.append(g.step_one_tick(34))
        dumps.append(g.dump())
    assert np.array_equal(np.array(eng, np.int32), z["eng_reward"])  # same run as the golden
    out[name] = np.array([d.encode() for d in dumps])
    for w, key in enumerate(("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul")):
        out[name + "__" + key] = g.durations(w)
np.savez_compressed(os.path.join(HERE, "dumps.npz"), **out)
print({k: (len(v), v.dtype) for k, v in out.items()})
print({k: v[:6] for k, v in out.items() if "__" in k and k.startswith("youturn_deaths")})
print(out["youturn_deaths"][120].decode())
