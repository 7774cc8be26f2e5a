"""The drop-in boundary without Python on top: a C++ program with nothing but HIP buffers and include/sfmi.h
(tests/capi/demo.cpp) is compiled against libsfmi.so, run, and must print the same numbers as the Python host
side fed with the same actions."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_caller_matches_python_host_side(tmp_path):
    from spacefortress_amd import SFVecEnv, _lib
    exe = str(tmp_path / "sfmi_demo")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["/opt/rocm/bin/hipcc", os.path.join(ROOT, "tests", "capi", "demo.cpp"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + libdir, "-lsfmi", "-Wl,-rpath," + libdir, "-o", exe])
    n, steps = 1000, 300
    out = subprocess.run([exe, str(n), str(steps)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = dict(kv.split("=") for kv in out.stdout.split())
    env = SFVecEnv(n, gametype="youturn", spawn_stride=1)
    env.reset()
    lcg, rsum = 12345, 0
    for t in range(steps):
        a = np.empty(n, np.uint8)
        for i in range(n):
            lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
            a[i] = (lcg >> 16) % 5
        obs, rew, done, info = env.step_tensors(torch.from_numpy(a).to(env.device))
        rsum += int(rew.sum())
    assert int(got["reward_sum"]) == rsum
    assert int(got["shots"]) == int(env.get_field("stats")[7].sum())
    assert float(got["ship_x_sum"]) == float(sum(env.get_field("ship_x").tolist()))  # same left-to-right double sum
    assert abs(float(got["obs_sum"]) - float(obs.cpu().numpy().astype(np.float64).sum())) < 1e-6 * abs(float(got["obs_sum"]))
    assert int(got["dim"]) == 19 and int(got["n_act"]) == 5
    env.close()


def test_steps_can_be_captured_in_a_hip_graph():
    """sf_step is a pure stream operation (no allocation, no synchronisation): K steps captured once in a HIP
    graph replay bit-identically to K direct launches."""
    from spacefortress_amd import SFVecEnv
    n, K = 4096, 32
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.randint(0, 5, (K, n), device="cuda", dtype=torch.uint8, generator=g)
    a, b = SFVecEnv(n, spawn_stride=1), SFVecEnv(n, spawn_stride=1)
    outs = [tuple(t.clone() for t in a.step_tensors(acts[k])) for k in range(K)]
    bufs = [(torch.empty(n, 19, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda"),
             torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")) for _ in range(K)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for k in range(K):
                b.step_tensors(acts[k], out=bufs[k])
    # capturing records the launches without running them: the state is still the initial one
    graph.replay()
    torch.cuda.synchronize()
    for k in range(K):
        for x, y in zip(outs[k], bufs[k]):
            assert torch.equal(x, y), k
    sa, sb = a.state_dict(), b.state_dict()
    assert all(np.array_equal(sa[f], sb[f]) for f in sa)
    a.close()
    b.close()


def test_new_entry_points_reject_bad_arguments():
    """Status codes of the entry points added around the path (image, stack, normaliser, trainer helpers)."""
    import ctypes as C
    from spacefortress_amd import SFVecEnv, _lib
    L = _lib.lib()
    env = SFVecEnv(8, obs_type="features")
    img = SFVecEnv(8, obs_type="image")
    buf = torch.zeros(8 * 4 * 84 * 84 + 64, dtype=torch.uint8, device=env.device)
    p = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
    assert L.sf_render(env._h, 0, p(buf), 0, None) == _lib.SF_ERR_ARG             # mode is not an image mode
    assert L.sf_render(env._h, 4, p(buf, 4), 0, None) == _lib.SF_ERR_ARG          # misaligned frames
    assert L.sf_render(env._h, 4, p(buf), 1000, None) == _lib.SF_ERR_ARG          # stride smaller than a frame
    assert L.sf_render(env._h, 4, p(buf), 0, None) == 0                           # any batch can be rendered
    assert L.sf_render_stack(img._h, p(buf), 4, 4, None, None) == _lib.SF_ERR_ARG  # slot out of range
    assert L.sf_render_stack(img._h, p(buf), 4, 3, None, None) == 0
    assert L.sf_set_event_output(env._h, p(buf, 2)) == _lib.SF_ERR_ARG
    a = torch.zeros(8, dtype=torch.uint8, device=env.device)
    assert L.sf_rollout(img._h, p(a), 1, 1, p(buf), None, None, None, None) == 0  # (with frames: step launches + frames, round 4)
    assert L.sf_step_record(env._h, p(a), 1, None, None, None, None, None, None, None, None, None, None) == _lib.SF_ERR_ARG
    assert L.sf_compute_returns(0, 8, None, None, None, None, None, 1, 0.99, 0.95, None) == _lib.SF_ERR_ARG
    z = C.c_void_p()
    from spacefortress_amd.vecnormalize import _Params
    bad = _Params(8, 40, 0, 0, 1, 1, 10., 10., .99, 1e-8)
    assert L.sf_normalizer_create(C.byref(bad), C.byref(z)) == _lib.SF_ERR_ARG     # obs_dim out of range
    ok = _Params(16, 19, 0, 0, 1, 1, 10., 10., .99, 1e-8)
    assert L.sf_normalizer_create(C.byref(ok), C.byref(z)) == 0
    r = torch.zeros(8, dtype=torch.int32, device=env.device)
    o = torch.zeros(8, 19, device=env.device)
    f = torch.zeros(8, device=env.device)
    d = torch.zeros(8, dtype=torch.uint8, device=env.device)
    # a normaliser made for another batch size
    assert L.sf_step_normalize(env._h, z, p(a), 1, p(o), p(r), p(d), p(d), p(f), 0, None) == _lib.SF_ERR_ARG
    assert L.sf_normalize(z, p(o), None, None, None, 0, None) == _lib.SF_ERR_ARG   # inputs and outputs come in pairs
    assert L.sf_normalizer_destroy(z) == 0
    torch.cuda.synchronize()
    env.close()
    img.close()


def test_the_shipped_binary_was_built_from_the_shipped_sources():
    """The GPU box runs a prebuilt libsfmi.so (git-ignored, it travels with the snapshot): its compiled-in build id must
    be the hash of the sources and flags that travelled with it."""
    from spacefortress_amd import _lib
    from spacefortress_amd import build as sfbuild

    assert _lib.lib().sf_build_id().decode() == sfbuild.source_hash()


def test_gym_namespace_builds_the_registered_ids():
    """`import spacefortress.gym`: the four ids the reference registers (obs_type 'image'), as device batches and as
    rl/envs.py:10-16's thunks; the symbolic observations as the explicit kwarg they are in the reference."""
    import spacefortress.gym as sfg
    for env_id, (n_act, dim) in {"SpaceFortress-youturn-image-v0": (5, 19), "SpaceFortress-autoturn-image-v0": (3, 17),
                                 "SpaceFortress-testyouturn-image-v0": (5, 19), "SpaceFortress-testautoturn-image-v0": (3, 17)}.items():
        v = sfg.make_vec_env(env_id, 6, obs_type="features")
        assert v.action_space.n == n_act and v.observation_space.shape == (dim,)
        o, r, d, i = v.step(np.zeros(6, np.int64))
        assert o.shape == (6, dim) and r.dtype == np.int64 and d.dtype == bool
        v.close()
        # the thunk of rl/envs.py:10-16: gym.make(id) -> seed -> WrapPyTorch: [1, 84, 84] uint8 frames
        e = sfg.make_env(env_id, 0, 0)()
        assert tuple(e.observation_space.shape) == (1, 84, 84) and e.action_space.n == n_act
        o = e.reset()
        assert o.shape == (1, 84, 84) and o.dtype == np.uint8 and o.max() > 100
        o, r, d, i = e.step(1)
        assert o.shape == (1, 84, 84) and isinstance(r, int) and d is False and i is False
        e.close()
    img = sfg.make_vec_env("SpaceFortress-youturn-image-v0", 4)  # default = the registered obs_type: what rl/envs.py wraps workers into
    assert img.observation_space.shape == (1, 84, 84)
    frames = img.reset()
    assert frames.shape == (4, 1, 84, 84)
    # the single-env thunk's frame is the batch's: same game, same INTER_AREA (host arithmetic == device arithmetic)
    e = sfg.make_env("SpaceFortress-youturn-image-v0", 0, 0)()
    one = sfg.make_vec_env("SpaceFortress-youturn-image-v0", 1, spawn_skip=0)
    assert np.array_equal(e.reset(), one.reset()[0].cpu().numpy())
    one.close()
    img.close()
    e.close()
    e = sfg.make_env("SpaceFortress-autoturn-image-v0", 0, 0, obs_type="features")()
    assert e.reset().shape == (17,) and e.action_space.n == 3
    e.close()
    # ENV:50-60: scale / viewport / ls are accepted (they only shape the picture); a picture in another geometry is not built
    e = sfg.SSF_Env(gametype="youturn", scale=1.0, viewport=(0, 0, 710, 626), ls=2, obs_type="features")
    assert (e.w, e.h) == (710, 626) and e.reset().shape == (19,)
    with pytest.raises(ValueError):
        e.render("rgb_array")
    e.close()
    with pytest.raises(ValueError):
        sfg.SSF_Env(gametype="youturn", scale=1.0, obs_type="image")
    with pytest.raises(KeyError):
        sfg.make_vec_env("SpaceFortress-nope-v0", 2)
