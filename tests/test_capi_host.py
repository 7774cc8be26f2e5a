"""CPU-only checks of libsfmi.so: it loads, exports every symbol include/sfmi.h declares, the
host-side tables equal what the reference engine produced (tests/golden/tables.npz), errors map
to the reference's exception classes, and -- with no GPU -- creating a batch fails loudly
instead of falling back to anything."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def L():
    from spacefortress_amd import build as sfbuild
    from spacefortress_amd import _lib

    sfbuild.build()
    return _lib.lib()


def test_exports_every_declared_symbol(L):
    from spacefortress_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "sfmi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sf_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(L, name), name
    assert L.sf_version() == 1


def test_binary_is_stamped_with_the_hash_of_its_sources(L):
    """sf_build_id() = sha256 over csrc/*, include/sfmi.h and the compiler flags (build.py: source_hash): the library
    that is loaded was built from the sources that lie next to it (tests/test_gpu_capi_native.py asserts the same on
    the GPU box, where the binary arrives prebuilt)."""
    from spacefortress_amd import build as sfbuild

    assert L.sf_build_id().decode() == sfbuild.source_hash() and len(sfbuild.source_hash()) == 16
    assert not sfbuild.needs_build()


def test_no_wide_buffer_store_without_its_wait_state(L):
    """The shipped binary, disassembled: no 128-bit buffer store with an SGPR soffset whose next instruction overwrites
    its data (build.py: scan_wide_store_hazard -- the compiler does not cover that form, and with more than one wave on a
    SIMD lanes 12-15 of every 16 stored the next instruction's result: batches beyond 65 536 envs, rounds 1-3).  The
    scanner itself is checked on a two-line sample of each kind."""
    from spacefortress_amd import _lib
    from spacefortress_amd import build as sfbuild

    bad = "\tbuffer_store_dwordx4 v[0:3], v6, s[56:59], s0 offen sc1\n\tv_lshlrev_b32_e32 v0, 2, v5\n"
    good = ["\tbuffer_store_dwordx4 v[0:3], v6, s[56:59], s0 offen sc1\n\ts_nop 1\n\tv_lshlrev_b32_e32 v0, 2, v5\n",
            "\tbuffer_store_dwordx4 v[0:3], v6, s[56:59], s0 offen sc1\n\tv_lshlrev_b32_e32 v4, 2, v5\n",
            "\tbuffer_store_dwordx4 v[0:3], v6, s[56:59], 0 offen\n\tv_lshlrev_b32_e32 v0, 2, v5\n",  # immediate: the compiler's case
            "\tbuffer_store_dwordx2 v[0:1], v6, s[56:59], s0 offen\n\tv_lshlrev_b32_e32 v0, 2, v5\n"]
    assert len(sfbuild.scan_wide_store_hazard(bad)) == 1
    assert len(sfbuild.scan_wide_store_hazard(bad.replace("v_lshlrev_b32_e32 v0", "v_add_f64 v[2:3], v[8:9]"))) == 1
    for g in good:
        assert sfbuild.scan_wide_store_hazard(g) == []
    text = sfbuild.device_disassembly(_lib.LIB_PATH)
    assert text.count("buffer_store_dwordx4") > 500 and "sf_render_kernel" in text and "sf_step_kernel" in text
    assert sfbuild.scan_wide_store_hazard(text) == []


def test_no_cpu_fallback_symbols(L):
    """The product library carries no CPU implementation of the path and never links the oracle."""
    import subprocess
    from spacefortress_amd import _lib

    syms = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    assert "sf_cpu" not in syms and "sfo_" not in syms
    needed = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH], text=True)
    assert "libamdhip64" in needed and "sforacle" not in needed
    src = "".join(open(os.path.join(ROOT, "spacefortress_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "spacefortress_amd")) if f.endswith(".py"))
    assert "oracle" not in src.replace("# oracle", "")


def test_host_tables_equal_the_reference(L):
    tab = np.load(os.path.join(GOLDEN, "tables.npz"))
    for seed in (1, 12345):
        sp = np.zeros((4096, 4), np.int16)
        assert L.sf_spawn_table(seed, 4096, sp.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(sp[:, :3], tab["spawns_seed%d" % seed]) and not sp[:, 3].any()
    tr = np.zeros((360, 2))
    assert L.sf_trig_table(tr.ctypes.data_as(C.c_void_p)) == 0
    # the reference's missile velocity / thrusted ship velocity at every integer heading
    assert (20 * tr).tobytes() == tab["missile_vel_by_angle"].tobytes()
    assert (tab["start_vel"][None, :] + 0.3 * tr).tobytes() == tab["thrust_vel_by_angle"].tobytes()
    hp = np.zeros(12)
    assert L.sf_hex_points(200, hp.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(hp, tab["hex_points"][:12])
    assert L.sf_hex_points(40, hp.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(hp, tab["hex_points"][12:])


def test_spawn_table_leaves_libc_rand_alone(L):
    libc = C.CDLL("libc.so.6")
    libc.srand(7)
    a = [libc.rand() for _ in range(5)]
    libc.srand(7)
    sp = np.zeros((64, 4), np.int16)
    L.sf_spawn_table(1, 64, sp.ctypes.data_as(C.c_void_p))
    assert [libc.rand() for _ in range(5)] == a


def test_presets_and_action_tables(L, oracle_mod):
    from spacefortress_amd import _lib
    from golden.make_golden import action_table

    p = _lib.Preset()
    for gt, auto, shaped in (("youturn", 0, 1), ("autoturn", 1, 1), ("test-youturn", 0, 0), ("test-autoturn", 1, 0)):
        assert L.sf_preset_get(gt.encode(), C.byref(p)) == 0
        assert (p.auto_turn, p.shaped, p.game_time, p.n_keys) == (auto, shaped, 180000, 2 if auto else 4)
        assert p.destroy_fortress == (1 if shaped else 100) and p.missile_penalty == (0.05 if shaped else 2.0)
        assert (p.width, p.height, p.big_hex, p.small_hex) == (710, 626, 200, 40)
        assert p.start_vx.hex() == "0x1.0000000000001p-1" and p.start_vy.hex() == "-0x1.bb67ae8584caap-1"
        for aset in (1, 0, -1):
            keys = (C.c_uint8 * 16)()
            n = L.sf_action_table(gt.encode(), aset, keys)
            assert list(keys[:n]) == action_table(gt, aset) == oracle_mod.OracleEnv(gt, action_set=aset).action_keys()
    assert L.sf_preset_get(b"nope", C.byref(p)) == _lib.SF_ERR_PRESET
    assert "Unknown config value" in _lib.last_error()  # SRC/pymodule.cpp:341
    keys = (C.c_uint8 * 16)()
    assert L.sf_action_table(b"youturn", 7, keys) == _lib.SF_ERR_ARG


def test_field_table(L):
    from spacefortress_amd import _lib

    d = _lib.FieldDesc()
    names = []
    for f in range(L.sf_n_fields()):
        assert L.sf_field_info(f, C.byref(d)) == 0
        names.append(d.name.decode())
        assert d.elem_size in (1, 2, 4, 8) and d.count in (1, 13, 20)
        assert L.sf_field_id(d.name) == f
    for want in ("ship_x", "ship_y", "ship_vx", "ship_vy", "ship_angle", "missile_x", "shell_vx", "stats",
                 "points", "raw_points", "vlner", "time", "prev_vlner", "spawn_cursor", "flags"):
        assert want in names
    assert L.sf_field_id(b"bogus") == _lib.SF_ERR_FIELD
    assert L.sf_field_info(999, C.byref(d)) == _lib.SF_ERR_FIELD


def test_create_fails_loudly_without_a_gpu(L):
    import torch
    from spacefortress_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    p = _lib.CreateParams(b"youturn", 16, 0, 1, 0, 0, 1, 0, 0, 0)
    h = C.c_void_p()
    assert L.sf_create(C.byref(p), C.byref(h)) == _lib.SF_ERR_NO_DEVICE and not h
    assert "no CPU path" in _lib.last_error()
    import spacefortress_amd

    with pytest.raises(_lib.SfmiError):
        spacefortress_amd.SFVecEnv(4)
    # argument errors come before the device is touched, with the reference's exception classes
    p = _lib.CreateParams(b"nope", 16, 0, 1, 0, 0, 1, 0, 0, 0)
    assert L.sf_create(C.byref(p), C.byref(h)) == _lib.SF_ERR_PRESET
    with pytest.raises(RuntimeError):
        _lib.check(_lib.SF_ERR_PRESET)
    for bad in (dict(n_envs=0), dict(obs_type=9), dict(action_set=3), dict(spawn_table_len=1000)):
        kw = dict(gametype=b"youturn", n_envs=16, device_id=0, action_set=1, obs_type=0, flags=0, seed=1,
                  spawn_skip=0, spawn_stride=0, spawn_table_len=0)
        kw.update(bad)
        assert L.sf_create(C.byref(_lib.CreateParams(**kw)), C.byref(h)) == _lib.SF_ERR_ARG, bad


def test_namespace_shim():
    import spacefortress.core as sf
    import spacefortress.gym as g

    assert (sf.FIRE_KEY, sf.THRUST_KEY, sf.LEFT_KEY, sf.RIGHT_KEY, sf.MAX_MISSILES, sf.MAX_SHELLS) == (1, 2, 3, 4, 20, 20)
    assert set(g.ENV_IDS) == {"SpaceFortress-%s-image-v0" % k for k in ("youturn", "autoturn", "testyouturn", "testautoturn")}
    assert callable(g.make_env("SpaceFortress-youturn-image-v0", 0, 0))
    with pytest.raises(KeyError):
        g.make_env("Pong-v0", 0, 0)()


def test_gym_registration_is_the_references(monkeypatch):
    """With gym importable, `import spacefortress.gym` makes the reference's four register() calls with the reference's
    kwargs -- obs_type 'image' on every id (python/spacefortress.gym/spacefortress/gym/__init__.py:3-29) -- so that
    rl/envs.py:10-16 (gym.make -> WrapPyTorch -> cv2.resize) receives a frame.  gym is not in this image: a stand-in
    `gym.envs.registration` records the calls."""
    import importlib
    import sys
    import types

    calls = []
    gym = types.ModuleType("gym")
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **kw: calls.append(kw)
    gym.envs, envs.registration = envs, reg
    for name, mod in (("gym", gym), ("gym.envs", envs), ("gym.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, name, mod)
    import spacefortress.gym as g

    importlib.reload(g)
    want = {"SpaceFortress-youturn-image-v0": "youturn", "SpaceFortress-autoturn-image-v0": "autoturn",
            "SpaceFortress-testyouturn-image-v0": "test-youturn", "SpaceFortress-testautoturn-image-v0": "test-autoturn"}
    assert len(calls) == 4 and {c["id"] for c in calls} == set(want)
    for c in calls:
        assert c["entry_point"] == "spacefortress.gym.envs:SSF_Env" and c["nondeterministic"] is False
        assert c["kwargs"] == {"gametype": want[c["id"]], "obs_type": "image"}
    # the entry point resolves to the class the thunk builds
    mod, cls = calls[0]["entry_point"].split(":")
    assert getattr(importlib.import_module(mod), cls) is g.SSF_Env
    monkeypatch.undo()
    importlib.reload(g)


def test_wrap_pytorch_is_inter_area_to_1x84x84():
    """rl/envs.py:19-30 on the host: [92, 90] grey frame -> [1, 84, 84] uint8, the library's INTER_AREA."""
    import spacefortress.gym as g
    sys.path.insert(0, ROOT)
    from oracle import render_np

    class Fake:
        action_space = "A"
        metadata = {"k": 1}

        def reset(self):
            return np.arange(92 * 90, dtype=np.uint32).reshape(92, 90).astype(np.uint8)

        def step(self, a):
            return self.reset()[::-1].copy(), 1, False, True

    w = g.WrapPyTorch(Fake())
    assert tuple(w.observation_space.shape) == (1, 84, 84) and w.observation_space.dtype == np.uint8
    o = w.reset()
    assert o.shape == (1, 84, 84) and o.dtype == np.uint8
    assert np.array_equal(o[0], render_np.resize_area(Fake().reset()))
    o2, r, d, i = w.step(0)
    assert o2.shape == (1, 84, 84) and (r, d, i) == (1, False, True) and w.action_space == "A"
    with pytest.raises(ValueError):
        w.observation(np.zeros(19))
