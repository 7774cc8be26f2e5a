"""The image observation on the GPU (sf_render behind SFVecEnv / SSF_Env) against (1) FRAMES DRAWN BY THE REFERENCE'S OWN
RENDERER -- SRC/draw.cpp + wireframe.cpp against cairo 1.16, tests/golden/frames, made by make_frames_golden.py -- on the
states they were drawn from, and (2) the model (oracle/render_np.py over oracle/cairo_model.c, itself bit-exact on those
fixtures and against the real cairo) on states recorded from the real reference engine (tests/golden/*.npz) and on oracle
lock-step runs.

Bar: BIT-EXACT on EVERY pixel, against the reference's frames and against the model -- the score text's rows included since
round 6: the text is drawn from a glyph atlas (sf_glyphs.h) taken from the image's cairo + FreeType and held to frames of the
reference's renderer (tests/golden/frames/scores.npz, make_score_golden.py).  The 84x84 image is OpenCV's INTER_AREA of that
surface (cv2 is not in this image: its published algorithm, restated twice)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    from spacefortress_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsfmi.so not built: the GPU tests never fall back"
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


@pytest.fixture(scope="module")
def model():
    from oracle import render_np as R

    z = np.load(os.path.join(GOLDEN, "tables.npz"))
    hb, hs = z["hex_points"][:12], z["hex_points"][12:]  # big, small: recorded from the reference
    return R, hb, hs, R.background(hb, hs)


def frames_close(got, want, what):
    """(the name is round 4's, when the kernel's anti-aliasing was a model of its own: now every pixel is equal)"""
    if not np.array_equal(got, want):
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        raise AssertionError((what, int(d.max()), int((d > 0).sum()), np.argwhere(d > 0)[:6].tolist()))


def _load_snaps(env, snaps, gametype="youturn"):
    """Put reference / oracle snapshots into the lanes of a batch (set_field) -- one snapshot per lane."""
    from sfcompare import snapshots_to_fields
    for k, v in snapshots_to_fields(snaps).items():
        env.set_field(k, v)


@pytest.mark.parametrize("name", ["poses.npz", "scenarios.npz"])
def test_frames_equal_the_references_own_renderer(sfa, name):
    """Every fixture frame: the state the reference drew it from goes into a lane (set_field), the frame kernel draws it,
    and every pixel of all 92 rows -- the score text too -- equals what SRC/draw.cpp + cairo 1.16 drew."""
    z = np.load(os.path.join(GOLDEN, "frames", name))
    frames, snaps, gt = z["frames"], z["snaps"], z["gametype"]
    rows = 0  # (rounds 1-5 masked meta["text_rows"] = 9 rows here)
    names = ["youturn", "autoturn", "test-youturn", "test-autoturn"]
    total = 0
    for g in sorted(set(int(x) for x in gt)):
        idx = np.flatnonzero(gt == g)
        env = sfa.SFVecEnv(len(idx), gametype=names[g].replace("test-", "test"), obs_type="image-raw") if False else \
            sfa.SFVecEnv(len(idx), gametype=names[g], obs_type="image-raw")
        _load_snaps(env, snaps[idx])
        got = env.render("image-raw").cpu().numpy()
        bad = [(str(z["labels"][i]), int(np.abs(got[k][rows:].astype(int) - frames[i][rows:].astype(int)).max()))
               for k, i in enumerate(idx) if not np.array_equal(got[k][rows:], frames[i][rows:])]
        assert not bad, (len(bad), bad[:8])
        total += len(idx)
        env.close()
    assert total == len(frames) > 600


def test_score_text_equals_the_references_own_renderer(sfa):
    """scores.npz: 2 200 scores drawn by the reference's renderer on a quiet state (rows 0..8) -- through the cached score
    pictures (|points| < 512), the in-place path (beyond) -- and 320 whole frames with explosion rings, ships, missiles and
    shells UNDER the text (the in-place path over objects, the explosion cache's score box).  Both frame sizes' code paths:
    the raw surface is compared, the 84x84 image against cv2's arithmetic on the reference's frame."""
    from oracle import render_np as R
    z = np.load(os.path.join(GOLDEN, "frames", "scores.npz"))
    pts, rows = z["points"], z["rows"]
    n = len(pts)
    base = np.repeat(z["base"].reshape(1), n)
    base["points"] = pts
    env = sfa.SFVecEnv(n, gametype="youturn", obs_type="image-raw")
    assert env.score_glyphs()["layout"] == (4, 4, 4, 1)
    _load_snaps(env, base)
    got = env.render("image-raw").cpu().numpy()
    bad = [int(pts[i]) for i in range(n) if not np.array_equal(got[i][:9], rows[i])]
    assert not bad, (len(bad), bad[:10])
    assert all(np.array_equal(got[i][9:], got[0][9:]) for i in range(0, n, 97))
    small = env.render("image").cpu().numpy()[:, 0]
    for i in range(0, n, 53):
        ref = got[0].copy()
        ref[:9] = rows[i]
        assert np.array_equal(small[i], R.resize_area(ref)), int(pts[i])
    env.close()
    frames, snaps = z["frames"], z["snaps"]
    env = sfa.SFVecEnv(len(snaps), gametype="youturn", obs_type="image-raw")
    _load_snaps(env, snaps)
    for rep in range(2):  # twice: the second frame of a dead ship comes out of the explosion cache, its score box with it
        got = env.render("image-raw").cpu().numpy()
        for i in range(len(snaps)):
            frames_close(got[i], frames[i], ("under the text", rep, i))
        small = env.render("image").cpu().numpy()[:, 0]
        for i in range(0, len(snaps), 7):
            frames_close(small[i], R.resize_area(frames[i]), ("under the text, 84x84", rep, i))
    env.close()


def test_score_glyphs_can_be_set_and_named_fallback(sfa, model):
    """sf_set_score_glyphs on the default geometry: the seven-segment fallback by name (equal to the model's fallback, NOT to
    the reference), an atlas of another 'font' (the built-in one moved a row down and thinned), and back to the built-in one
    through set_image_geometry -- cached pictures, baked backgrounds and in-place text all follow."""
    R, hb, hs, bg = model
    z = np.load(os.path.join(GOLDEN, "frames", "scores.npz"))
    pts = np.array([0, 7, 123, 511, 512, 99999, -3], np.int32)
    base = np.repeat(z["base"].reshape(1), len(pts))
    base["points"] = pts
    env = sfa.SFVecEnv(len(pts), gametype="youturn", obs_type="image-raw")
    _load_snaps(env, base)
    builtin = env.score_glyphs()
    f0 = env.render("image-raw").cpu().numpy()
    for i, p in enumerate(pts):
        frames_close(f0[i], R.render_raw(base[i], hb, hs), ("built-in", int(p)))
    env.set_score_glyphs(None)
    assert env.score_glyphs() is None
    f1 = env.render("image-raw").cpu().numpy()
    for i, p in enumerate(pts):
        frames_close(f1[i], R.render_raw(base[i], hb, hs, text="segments"), ("fallback", int(p)))
    assert np.abs(f1.astype(int) - f0.astype(int))[:, :9].max() > 50
    other = dict(alpha=(builtin["alpha"] // 2).astype(np.uint8), layout=(4, 4, 4, 2), x0=np.full((11, 10), 31, np.int16))
    env.set_score_glyphs(other["alpha"], other["layout"], other["x0"])
    f2 = env.render("image-raw").cpu().numpy()
    for i, p in enumerate(pts):
        frames_close(f2[i], R.render_raw(base[i], hb, hs, glyphs=other), ("another atlas", int(p)))
    with pytest.raises(ValueError):  # ink outside the default geometry's text box
        env.set_score_glyphs(builtin["alpha"], (4, 4, 4, 3), 31)
    env.set_image_geometry()  # the default geometry again: the built-in atlas
    assert np.array_equal(env.score_glyphs()["alpha"], builtin["alpha"])
    frames_close(env.render("image-raw").cpu().numpy(), f0, "built-in again")
    env.close()


def test_frames_equal_the_references_renderer_in_other_geometries(sfa):
    """sf_render_generic.hip against the reference's frames in four other geometries (one with vw * scale not whole) -- every
    row: each geometry's glyph atlas (score_glyphs.npz, from the same cairo) goes in through sf_set_score_glyphs; without one
    the text is the seven-segment fallback and only the rows below it are the reference's."""
    from oracle import render_np as R
    z = np.load(os.path.join(GOLDEN, "frames", "geometries.npz"))
    snaps = z["snaps"]
    for gi, (sc, vx, vy, vw, vh, ls) in enumerate(z["geometries"]):
        env = sfa.SFVecEnv(len(snaps), gametype="youturn", obs_type="image-raw", image_geometry=(sc, (vx, vy, vw, vh), ls))
        assert env.score_glyphs() is None and not env.default_geometry
        _load_snaps(env, snaps)
        want = z["frames_%d" % gi]
        got = env.render("image-raw").cpu().numpy()
        assert got.shape == want.shape
        rows = int((112 - vy) * (want.shape[1] / vh)) + 1
        for i in range(len(snaps)):
            frames_close(got[i][rows:], want[i][rows:], ("geometry, fallback text", gi, i))
        A = R.load_glyphs(gi + 1)
        env.set_score_glyphs(A["alpha"], A["layout"], A["x0"])
        got = env.render("image-raw").cpu().numpy()
        for i in range(len(snaps)):
            frames_close(got[i], want[i], ("geometry", gi, i))
        env.close()


def test_frames_equal_the_references_renderer_in_a_close_up(sfa):
    """SSF_Env(scale=0.75, viewport=(230, 185, 250, 260)): a 187 x 195 surface around the fortress (zoom.npz, drawn by the
    reference's renderer), the largest scale sf_set_image_geometry takes -- a wireframe's box is larger than one pass of the
    rasteriser's accumulators (windows of rows), the dead ship's arcs reach 47 pixels out, the big hexagon lies outside the
    view, no text rows: every pixel."""
    z = np.load(os.path.join(GOLDEN, "frames", "zoom.npz"))
    sc, vx, vy, vw, vh, ls = z["geometry"]
    snaps = z["snaps"]
    env = sfa.SFVecEnv(len(snaps), gametype="youturn", obs_type="image-raw", image_geometry=(sc, (vx, vy, vw, vh), ls))
    _load_snaps(env, snaps)
    got = env.render("image-raw").cpu().numpy()
    assert got.shape == z["frames"].shape == (len(snaps), 195, 187)
    for i in range(len(snaps)):
        frames_close(got[i], z["frames"][i], ("zoom", i))
    env.close()


@pytest.mark.parametrize("name,stride", [
    ("autoturn_destroy", 7), ("youturn_hunter", 61), ("youturn_deaths", 13), ("youturn_rapid_fire", 11),
    ("youturn_random_short", 9),
])
def test_frames_vs_model_on_reference_states(sfa, model, name, stride):
    """Replay a recorded reference run on the GPU; at every `stride`-th tick render both sizes and
    compare with the model's rendering of the state the REFERENCE had at that tick."""
    R, hb, hs, bg = model
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    assert meta["snap_every"] == 1
    N = 2
    env = sfa.SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"],
                       spawn_skip=meta["spawn_skip"], obs_type="image")
    acts = torch.from_numpy(np.repeat(z["actions"][:, None], N, 1).astype(np.uint8)).to(env.device)
    done = z["done"].astype(bool)
    # no reset(): the recording starts from the first Game of the process, which sf_create made
    obs0 = env.render("image")
    assert obs0.shape == (N, 1, 84, 84) and obs0.dtype == torch.uint8
    seen = dict(explosion=0, missiles=0, shells=0, kill_bar=0)
    T = min(len(acts), 1500)
    for t in range(T):
        obs, rew, dn, info = env.step_tensors(acts[t])
        if t % stride or done[t]:
            continue
        snap = z["snaps"][t]
        raw = env.render("image-raw").cpu().numpy()
        small = obs.cpu().numpy()
        want_raw = R.render_raw(snap, hb, hs, bg=bg)
        frames_close(raw[0], want_raw, (name, t, "raw"))
        assert np.array_equal(raw[0], raw[1])
        frames_close(small[0, 0], R.resize_area(want_raw), (name, t, "84"))
        assert np.array_equal(small[0, 0], R.resize_area(raw[0])), (name, t)  # the shrink itself: the kernel's own raw frame
        seen["explosion"] += int(not snap["ship_alive"] or not snap["fort_alive"])
        seen["missiles"] += int(snap["missile_alive"].sum() > 0)
        seen["shells"] += int(snap["shell_alive"].sum() > 0)
        seen["kill_bar"] += int(snap["vlner"] > 10)
    env.close()
    if name == "autoturn_destroy":
        assert seen["explosion"] and seen["missiles"] and seen["kill_bar"], seen
    if name == "youturn_deaths":
        assert seen["explosion"] and seen["shells"], seen


def test_frames_vs_model_random_lanes(sfa, oracle_mod, model):
    """Different lanes in different states (spawn offsets, random actions), lock-step with the CPU
    oracle; every lane's frame at a few checkpoints."""
    R, hb, hs, bg = model
    O = oracle_mod
    N, T = 48, 420
    rng = np.random.default_rng(11)
    env = sfa.SFVecEnv(N, gametype="youturn", obs_type="image-raw", spawn_stride=5, spawn_skip=1)
    orc = O.OracleVecEnv("youturn", N, spawn_stride=5, spawn_skip=1)
    acts = rng.integers(0, 5, (T, N)).astype(np.uint8)
    a = torch.from_numpy(acts).to(env.device)
    obs = env.reset()
    orc.reset()
    assert obs.shape == (N, 92, 90)
    for t in range(T):
        obs, rew, dn, info = env.step_tensors(a[t])
        orc.step(acts[t].astype(np.int32))
        if t % 70 == 69:
            got = obs.cpu().numpy()
            snaps = orc.snapshots()
            for i in range(N):
                frames_close(got[i], R.render_raw(snaps[i], hb, hs, bg=bg), (t, i))
    env.close()


def test_ship_lights_up_where_the_state_says(sfa, model):
    """Geometry, independent of the model: the pixels that differ from the static background and are
    not text / bar / fortress have their centroid at the ship's device position."""
    R, hb, hs, bg = model
    N = 256
    env = sfa.SFVecEnv(N, gametype="youturn", obs_type="features", spawn_stride=1)
    env.reset()
    a = torch.full((N,), 0, dtype=torch.uint8, device=env.device)
    for _ in range(3):
        env.step_tensors(a)
    raw = env.render("image-raw").cpu().numpy().astype(np.int32)
    sx, sy = env.get_field("ship_x"), env.get_field("ship_y")
    ys, xs = np.mgrid[0:92, 0:90]
    checked = 0
    for i in range(N):
        ex, ey = (sx[i] - 130) * 0.2, (sy[i] - 80) * 0.2
        if abs(ex - 45) < 13 and abs(ey - 47) < 13:
            continue  # too close to the fortress to tell their pixels apart
        checked += 1
        diff = np.abs(raw[i] - bg.astype(np.int32))
        diff[:8] = 0          # score text
        diff[86:] = 0         # vulnerability bar
        diff[41:54, 38:54] = 0  # fortress wireframe
        assert diff.sum() > 0
        cx, cy = (diff * (xs + 0.5)).sum() / diff.sum(), (diff * (ys + 0.5)).sum() / diff.sum()
        # the wireframe's centroid sits within its own extent (36 user units = 7 px) of the position
        assert abs(cx - ex) < 3.0 and abs(cy - ey) < 3.0, (i, cx, cy, ex, ey)
    assert checked > N // 2
    env.close()


def test_image_env_surfaces(sfa):
    """SSF_Env(obs_type='image') returns the [92, 90] grey frame, declares (92, 90, 3) like ENV:169,
    render('rgb_array') replicates it; the vec env's 'image' observation is [N, 1, 84, 84] uint8 with
    the WrapPyTorch Box (rl/envs.py:22-26); rollout refuses image observations."""
    e = sfa.SSF_Env(gametype="autoturn", obs_type="image")
    f0 = e.reset()
    assert f0.shape == (92, 90) and f0.dtype == np.uint8
    assert e.observation_space.shape == (92, 90, 3)
    f1, r, d, i = e.step(1)
    assert f1.shape == (92, 90) and isinstance(r, int) and isinstance(d, bool) and isinstance(i, bool)
    rgb = e.render("rgb_array")
    assert rgb.shape == (92, 90, 3) and np.array_equal(rgb[:, :, 0], f1) and np.array_equal(rgb[:, :, 2], f1)
    with pytest.raises(NotImplementedError):
        e.render("human")
    e.close()
    with pytest.raises(ValueError):
        sfa.SSF_Env(scale=.5)

    v = sfa.SFVecEnv(5, obs_type="image")
    assert v.observation_space.shape == (1, 84, 84) and v.observation_space.dtype == np.uint8
    o, r, d, i = v.step(np.zeros(5, np.int64))
    assert o.shape == (5, 1, 84, 84) and o.dtype == np.uint8 and r.dtype == np.int64
    assert np.array_equal(o, v.render("image").cpu().numpy())
    ob, rw, dn, inf = v.rollout(torch.zeros((4, 5), dtype=torch.uint8, device=v.device), want_obs=False)
    assert ob is None and rw.shape == (4, 5)
    v.close()
    # a rollout WITH frames: K step launches each followed by its frames, in the one call = K step_tensors calls
    rng = np.random.default_rng(2)
    acts = torch.from_numpy(rng.integers(0, 5, (40, 192)).astype(np.uint8)).cuda()
    va, vb = sfa.SFVecEnv(192, obs_type="image", spawn_stride=1), sfa.SFVecEnv(192, obs_type="image", spawn_stride=1)
    va.reset()
    vb.reset()
    ob, rw, dn, inf = va.rollout(acts)
    assert ob.shape == (40, 192, 1, 84, 84) and ob.dtype == torch.uint8
    for t in range(40):
        o, r, d, i = vb.step_tensors(acts[t])
        assert torch.equal(ob[t], o) and torch.equal(rw[t], r) and torch.equal(dn[t], d) and torch.equal(inf[t], i), t
    assert torch.equal(va.render("image"), vb.render("image")) and ob[-1].max() > 0
    for k, x in va.state_dict().items():
        assert np.array_equal(x, vb.state_dict()[k]), k
    # ... and on actions drawn in the kernel, with the event masks of every step: the same stream as single sampled steps
    va.seed_actions(9)
    vb.seed_actions(9)
    va.enable_events()
    vb.enable_events()
    ob, rw, dn, inf, pa = va.rollout_sampled(12)
    ev = va.rollout_events  # [12, N]: one row of masks per tick
    assert ev is not None and tuple(ev.shape) == (12, 192)
    for t in range(12):
        pb = torch.empty(192, dtype=torch.uint8, device=vb.device)
        o, r, d, i = vb.step_sampled(actions_out=pb)
        assert torch.equal(ob[t], o) and torch.equal(rw[t], r) and torch.equal(pa[t], pb), t
        assert torch.equal(ev[t], vb.events), t
    va.close()
    vb.close()


def test_image_full_batch_properties(sfa):
    """65 536 envs: identical lanes give identical frames (one wave per env, no cross-talk), the
    static background is where it should be in every frame, and frames follow the state through
    the auto-reset at the end of the episode."""
    N = 65536
    env = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", reuse_buffers=True)
    rng = np.random.default_rng(5)
    obs = env.reset()
    first = obs[0].clone()
    assert bool((obs == first).all())
    for t in range(40):
        a = torch.full((N,), int(rng.integers(0, 5)), dtype=torch.uint8, device=env.device)
        obs, rew, dn, info = env.step_tensors(a)
    assert bool((obs == obs[0]).all()) and not bool((obs[0] == first).all())
    # different actions per lane from here on: frames must now differ between lanes, yet every
    # frame keeps the hexagon pixels (nothing ever draws darker than the white hexagon)
    a = torch.from_numpy(rng.integers(0, 5, (30, N)).astype(np.uint8)).to(env.device)
    for t in range(30):
        obs, rew, dn, info = env.step_tensors(a[t])
    assert int((obs != obs[0]).any(dim=(1, 2, 3)).sum()) > N // 2
    # a 0.6-pixel white stroke shrunk by INTER_AREA peaks around 140; explosions (.5 / .75 grey)
    # may be drawn over it, nothing darker
    import ctypes as C
    from spacefortress_amd import _lib
    bg, bg84 = np.zeros((92, 90), np.uint8), np.zeros((84, 84), np.uint8)
    _lib.lib().sf_image_background(bg.ctypes.data_as(C.c_void_p))
    _lib.lib().sf_resize_area_u8(bg.ctypes.data_as(C.c_void_p), 90, 92, bg84.ctypes.data_as(C.c_void_p), 84, 84)
    hexmask = torch.from_numpy(bg84 >= 100).to(env.device)
    assert int(hexmask.sum()) > 100
    assert bool((obs[:, 0][:, hexmask] >= 45).all())
    env.close()


def test_frame_stack_matches_the_trainers_shift(sfa):
    """FrameStack (device ring, frames rendered straight into a slot) against the trainer's own
    update: shift by one frame, zero finished envs, newest last (rl/train.py:51-56,92-97)."""
    N, S = 64, 4
    # episodes end at step 5295 in every lane; start late in the episode so that a reset is inside
    env = sfa.SFVecEnv(N, gametype="autoturn", obs_type="image", spawn_stride=2)
    twin = sfa.SFVecEnv(N, gametype="autoturn", obs_type="image", spawn_stride=2)
    for e in (env, twin):
        e.set_field("time", np.full(N, 34 * 5285, np.int32))
    fs = sfa.FrameStack(env, S)
    rng = np.random.default_rng(2)
    ref = torch.zeros((N, S, 84, 84), dtype=torch.uint8, device=env.device)
    # no reset(): start both from the state just set
    env.render("image", out=fs.ring[:, fs.head:fs.head + 1])
    ref[:, -1:] = twin.render("image")
    assert torch.equal(fs.stacked(), ref)
    saw_done = False
    for t in range(25):
        a = torch.from_numpy(rng.integers(0, 3, N).astype(np.uint8)).to(env.device)
        rew, done, info = fs.step(a)
        obs, r2, d2, i2 = twin.step_tensors(a)
        assert torch.equal(rew, r2) and torch.equal(done, d2.bool())
        ref *= (1 - d2)[:, None, None, None]
        ref[:, :-1] = ref[:, 1:].clone()
        ref[:, -1:] = obs
        assert torch.equal(fs.stacked(), ref), t
        saw_done = saw_done or bool(done.any())
    assert saw_done
    # a stack straight after reset(): three empty frames and the first observation
    st = fs.reset()
    assert st.shape == (N, S, 84, 84) and int(st[:, :-1].sum()) == 0 and int(st[:, -1].sum()) > 0
    env.close()
    twin.close()


def test_explosion_cache_is_invisible(sfa, monkeypatch):
    """A dead ship's explosion is drawn once and then restored from the per-env cache for the
    rest of its 30 frames; frames must be identical to a batch that redraws it every time,
    in both sizes, through deaths, respawns and a restored state."""
    N, T = 512, 260
    rng = np.random.default_rng(9)
    acts = torch.from_numpy(rng.integers(0, 5, (T, N)).astype(np.uint8)).cuda()
    monkeypatch.setenv("SFMI_NO_EXPLOSION_CACHE", "1")
    plain = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=3)
    plain.render("image")  # the switch is read by a batch's first frame
    monkeypatch.delenv("SFMI_NO_EXPLOSION_CACHE")
    cached = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=3)
    dead_frames = 0
    for t in range(T):
        o1, *_ = plain.step_tensors(acts[t])
        o2, *_ = cached.step_tensors(acts[t])
        assert torch.equal(o1, o2), t
        if t % 16 == 0:
            assert torch.equal(plain.render("image-raw"), cached.render("image-raw")), t
        dead_frames += int((torch.from_numpy(cached.get_field("flags")) & 1 == 0).sum()) if t % 16 == 0 else 0
    assert dead_frames > 100
    # a state restored into the batch: the key is the position, stale entries cannot match by accident
    sd = plain.state_dict()
    perm = rng.permutation(N)
    for e in (plain, cached):
        e.load_state_dict({k: (v[..., perm] if v.ndim else v) for k, v in sd.items()})
    assert torch.equal(plain.render("image"), cached.render("image"))
    assert torch.equal(plain.render("image-raw"), cached.render("image-raw"))
    plain.close()
    cached.close()


def test_fortress_explosion_picture_is_invisible(sfa, monkeypatch):
    """The destroyed fortress's explosion is one picture per batch, restored while the ship's pixels stay clear of it
    and drawn in place otherwise.  Replaying the recorded run that destroys the fortress (twice), a batch with the
    pictures and one that draws everything in place give identical frames in both sizes at every tick."""
    z = np.load(os.path.join(GOLDEN, "autoturn_destroy.npz"))
    meta = json.loads(str(z["meta"]))
    N = 3
    mk = lambda: sfa.SFVecEnv(N, gametype=meta["gametype"], action_set=meta["action_set"], seed=meta["seed"],
                              spawn_skip=meta["spawn_skip"], obs_type="image")
    monkeypatch.setenv("SFMI_NO_EXPLOSION_CACHE", "1")
    plain = mk()
    plain.render("image")  # the switch is read by a batch's first frame
    monkeypatch.delenv("SFMI_NO_EXPLOSION_CACHE")
    cached = mk()
    acts = torch.from_numpy(np.repeat(z["actions"][:, None], N, 1).astype(np.uint8)).cuda()
    dead = 0
    for t in range(len(acts)):
        o1, *_ = plain.step_tensors(acts[t])
        o2, *_ = cached.step_tensors(acts[t])
        assert torch.equal(o1, o2), t
        fort_dead = int((torch.from_numpy(cached.get_field("flags")) & 2 == 0).sum())
        dead += fort_dead
        if fort_dead:
            assert torch.equal(plain.render("image-raw"), cached.render("image-raw")), t
    assert dead >= 20 * N, dead
    plain.close()
    cached.close()


@pytest.mark.parametrize("gametype,policy,N,T", [("test-autoturn", "hunter", 1024, 700), ("youturn", "random", 2048, 400)])
def test_every_shortcut_of_the_render_kernel_is_invisible(sfa, monkeypatch, gametype, policy, N, T):
    """tools/render_soak.py in small: the product batch (explosion cache with the score / bar boxes under an explosion, the
    pictures drawn once per batch -- fortress headings, fortress explosion, scores, bar states --, launch-order words, exact
    dirty boxes) against a batch that draws every frame in place in env order, same actions: every 84x84 frame identical
    at every step, the 92x90 ones on sampled steps.  The firing pattern of the first case destroys the fortress hundreds
    of times: scores up to the hundreds, every state of the bar, the fortress's explosion."""
    monkeypatch.setenv("SFMI_NO_EXPLOSION_CACHE", "1")
    monkeypatch.setenv("SFMI_NO_RENDER_ORDER", "1")
    plain = sfa.SFVecEnv(N, gametype=gametype, obs_type="image", spawn_stride=3)
    plain.render("image")  # the switches are read at create / by the first frame
    monkeypatch.delenv("SFMI_NO_EXPLOSION_CACHE")
    monkeypatch.delenv("SFMI_NO_RENDER_ORDER")
    prod = sfa.SFVecEnv(N, gametype=gametype, obs_type="image", spawn_stride=3)
    rng = np.random.default_rng(5)
    phase = rng.integers(0, 96, N)
    pat = np.array(([1] + [0] * 7) * 11 + [1, 0, 1, 0] + [0] * 4, np.uint8)  # (tools/soak.py: hunter)
    scores, bars, fort_dead = set(), set(), 0
    for t in range(T):
        acts = rng.integers(0, prod.n_actions, N).astype(np.uint8)
        if policy == "hunter":
            acts = np.where(rng.random(N) < 0.1, acts, pat[(t + phase) % len(pat)]).astype(np.uint8)
        a = torch.from_numpy(acts).cuda()
        o1, *_ = plain.step_tensors(a)
        o2, *_ = prod.step_tensors(a)
        assert torch.equal(o1, o2), (t, (o1 != o2).flatten(1).any(1).nonzero().flatten()[:8].tolist())
        if t % 16 == 0:
            assert torch.equal(plain.render("image-raw"), prod.render("image-raw")), t
        if t % 50 == 0:
            scores.update(np.unique(prod.get_field("points").astype(np.int64)).tolist())
            bars.update(np.unique(np.minimum(prod.get_field("vlner"), 11)).tolist())
            fort_dead += int(((prod.get_field("flags").astype(np.int64) & 2) == 0).sum())
    if policy == "hunter":
        assert len(scores) > 20 and max(scores) >= 100 and fort_dead > 100, (sorted(scores)[-3:], fort_dead)
        assert bars >= set(range(0, 12)), bars
    else:
        assert len(bars) >= 3, bars
    plain.close()
    prod.close()


def test_scores_beyond_the_picture_table_and_under_an_explosion(sfa, monkeypatch):
    """the equality soaks only saw scores 0..264.  The score has a picture for -512 <= points < 512; beyond that,
    and for negative points, it is drawn in place -- also when a dead ship's explosion lies under the score box (a ship lost
    through the top of the big hexagon), where the box is kept in the env's explosion-cache entry keyed by the points.  States
    with points on, around and far beyond the table's ends, half of the ships freshly dead right under the score: the
    product batch against one that draws everything in place, 40 ticks (the whole explosion and the respawn), both sizes."""
    N, T = 384, 40
    pts = np.array([-100000, -600, -513, -512, -511, -1, 0, 1, 7, 511, 512, 513, 600, 9999999, 1234567, 264], np.float32)
    monkeypatch.setenv("SFMI_NO_EXPLOSION_CACHE", "1")
    monkeypatch.setenv("SFMI_NO_RENDER_ORDER", "1")
    plain = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=3)
    plain.render("image")  # the switches are read at create / by the first frame
    monkeypatch.delenv("SFMI_NO_EXPLOSION_CACHE")
    monkeypatch.delenv("SFMI_NO_RENDER_ORDER")
    prod = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=3)
    rng = np.random.default_rng(17)
    for e in (plain, prod):
        e.reset()
    sd = prod.state_dict()
    sd["points"] = pts[np.arange(N) % len(pts)]
    sd["raw_points"] = sd["points"].copy()
    under = (np.arange(N) // len(pts)) % 2 == 1  # every other group: the ship died a moment ago, right under the score
    sd["ship_x"] = np.where(under, 355 + rng.integers(-60, 61, N), sd["ship_x"]).astype(np.float64)
    sd["ship_y"] = np.where(under, 150 + rng.integers(0, 12, N), sd["ship_y"]).astype(np.float64)
    sd["flags"] = np.where(under, sd["flags"] & ~np.uint8(1), sd["flags"]).astype(np.uint8)
    sd["ship_death_timer"] = np.where(under, 34 * rng.integers(0, 5, N), sd["ship_death_timer"]).astype(np.int32)
    for e in (plain, prod):
        e.load_state_dict(sd)
    assert torch.equal(plain.render("image"), prod.render("image"))
    assert torch.equal(plain.render("image-raw"), prod.render("image-raw"))
    lit = plain.render("image-raw")[:, 2:8, 30:60].flatten(1).float().mean(1).cpu().numpy()  # the score's rows: digits everywhere
    assert (lit > 5).all()
    for t in range(T):
        a = torch.from_numpy(rng.integers(0, 5, N).astype(np.uint8)).cuda()
        o1, *_ = plain.step_tensors(a)
        o2, *_ = prod.step_tensors(a)
        assert torch.equal(o1, o2), t
        if t % 4 == 0:
            assert torch.equal(plain.render("image-raw"), prod.render("image-raw")), t
    assert (prod.get_field("flags") & 1).sum() > N // 2  # the explosions ran their course: ships are back
    plain.close()
    prod.close()


def test_launch_order_hint_is_invisible(sfa, monkeypatch):
    """The step kernel tells the render launch which ships just died, and those frames start first (sf_render.hip:
    pick_env).  The words decide only when a frame is drawn: a batch without them, the batch's own, and every pattern
    written over them -- none, all, random at several densities, more marked envs than the front of the grid holds,
    a ragged last tile -- give the same frames, every env drawn exactly once (the canvas is pre-filled with a value no
    frame is made of whole, so an env nobody drew shows)."""
    import ctypes as C
    from spacefortress_amd import _lib
    N, T = 1000, 140  # not a multiple of 64: the last tile is ragged
    rng = np.random.default_rng(31)
    acts = torch.from_numpy(rng.integers(0, 5, (T, N)).astype(np.uint8)).cuda()
    monkeypatch.setenv("SFMI_NO_RENDER_ORDER", "1")
    plain = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=5)
    monkeypatch.delenv("SFMI_NO_RENDER_ORDER")
    hinted = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=5)
    L = _lib.lib()
    words = (N + 63) // 64
    assert L.sf_set_render_order_hint(plain._h, (C.c_uint64 * words)(), words) == _lib.SF_ERR_ARG  # no words to write
    deaths = 0
    for t in range(T):
        o1, *_ = plain.step_tensors(acts[t])
        o2, *_ = hinted.step_tensors(acts[t])
        assert torch.equal(o1, o2), t
        deaths += int((torch.from_numpy(hinted.get_field("flags")) & 1 == 0).sum())
    assert deaths > 50
    want84, want_raw = plain.render("image"), plain.render("image-raw")
    valid = np.zeros(words * 64, bool)
    valid[:N] = True
    for density in (0.0, 1.0, 0.01, 0.05, 0.3, 0.9):
        bits = (rng.random(words * 64) < density) & valid  # (the step kernel never marks a lane behind the batch)
        w = np.packbits(bits.reshape(words, 64), axis=1, bitorder="little").view(np.uint64).reshape(words)
        assert int(bits.sum()) == sum(bin(int(x)).count("1") for x in w)
        _lib.check(L.sf_set_render_order_hint(hinted._h, w.ctypes.data_as(C.c_void_p), words))
        for mode, want in (("image", want84), ("image-raw", want_raw)):
            out = torch.full_like(want, 7)
            got = hinted.render(mode, out=out)
            assert torch.equal(got, want), (density, mode)
    plain.close()
    hinted.close()


def test_config5_at_its_size_against_the_model(sfa, oracle_mod, model):
    """BASELINE.json configs[4] as written: youturn, 16 384 envs, the 84x84 grey raster with the trainer's 4-frame stack
    on the device.  160 random steps through FrameStack; on sampled lanes (the first, the last, some in between) the
    newest frame of the stack is compared with the numpy model's render of the oracle's state of that lane (90x92
    surface -> INTER_AREA), and the older slots with the frames of the previous steps."""
    R, hb, hs, bg = model
    O = oracle_mod
    N, S, T = 16384, 4, 160
    rng = np.random.default_rng(23)
    pick = np.sort(rng.choice(N, 24, replace=False))
    pick[0], pick[-1] = 0, N - 1
    ring = rng.integers(0, 5, (64, N)).astype(np.uint8)
    env = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=1, reuse_buffers=True)
    orcs = [O.OracleVecEnv("youturn", 1, spawn_skip=int(l)) for l in pick]
    fs = sfa.FrameStack(env, S)
    st = fs.reset()
    assert st.shape == (N, S, 84, 84) and st.dtype == torch.uint8
    for o in orcs:
        o.reset()
    dring = torch.from_numpy(ring).to(env.device)
    dpick = torch.from_numpy(pick).to(env.device)
    hist = []  # the model's 84x84 frames of the sampled lanes, newest last
    for t in range(T):
        rew, done, info = fs.step(dring[t % 64])
        frames = []
        for j, o in enumerate(orcs):
            o.step(ring[t % 64, pick[j]:pick[j] + 1].astype(np.int32))
            if t >= T - S:
                frames.append(R.resize_area(R.render_raw(o.snapshots()[0], hb, hs, bg=bg)))
        if t >= T - S:
            hist.append(np.stack(frames))
    got = fs.stacked()[dpick].cpu().numpy()  # [picked, S, 84, 84], oldest first
    for k in range(S):
        for j in range(len(pick)):
            frames_close(got[j, k], hist[k][j], ("slot", k, "lane", int(pick[j])))
    assert not bool(done.any())
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("text,name", [("segments", "render_fingerprints_youturn_1024x640_hunter.txt"),
                                       ("atlas", "render_fingerprints_r6_atlas_youturn_1024x640_hunter.txt")])
def test_frames_reproduce_the_recorded_fingerprints(sfa, text, name):
    """A fingerprint file holds a 63-bit weighted sum of every byte of every frame (84x84 each step, the raw 90x92 every
    eighth) of 1 024 envs over 640 steps of the fortress-hunting policy -- ships exploding, missiles, shells, the fortress
    destroyed, scores and every state of the bar --, made by tools/render_hash.py.  A single changed byte in 655 360 frames
    shows.
      "segments": round 5's file, made once its kernel equalled every frame of the reference's renderer outside the text
      rows (two independent lane arrangements of the scan -- commit 0a3dd54's general one and the frame kernel's -- gave the
      same 640 lines).  With the seven-segment fallback selected BY NAME today's kernel still has to give these sums: nothing
      but the text changed in round 6.
      "atlas": the same run with the built-in glyph atlas (the reference's text), recorded in round 6 under the build that
      passed every fixture test of this file with all 92 rows compared; the guard for every later change of the kernel.
    Both files were recorded BEFORE the explosion pre-pass existed (sf_explosion_kernel: batches up to 6 144 envs -- this run's 1 024
    -- have a freshly dead ship's explosion drawn by eight waves into its cache entry ahead of the frame kernel): that today's sums
    equal them is the test that the pre-pass fills an entry exactly as drawing in place did, over the 7 000 deaths of the run."""
    import sys

    from conftest import ROOT

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from render_hash import fingerprints

    want = open(os.path.join(GOLDEN, name)).read().split("\n")[:-1]
    got = fingerprints("youturn", 1024, 640, "hunter", text)
    assert len(got) == len(want) == 640
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, "frames differ from the recorded ones at steps %s ..." % bad[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("gametype,seed", [("youturn", 21), ("autoturn", 22)])
def test_crowded_constructed_frames_vs_model(sfa, oracle_mod, model, monkeypatch, gametype, seed):
    """Frames play never produces: up to 20 missiles and 20 shells anywhere on (and off) the screen, ships and explosions
    anywhere, a dead or live fortress at any heading (test_gpu_parity._fuzz_base).  They take the render kernel's rare paths
    -- the twentieth missile's own call, shells in slots 16..19, shells that do and do not fit the free top lanes of the
    missiles' range, several chunks of strokes, a fortress drawn in place under a ship, projectiles over the score and the
    bar (the restart from another background) -- against the numpy model, in both sizes; and with every shortcut switched
    off (drawn in place, env order) the frames must be the same bytes."""
    from test_gpu_parity import _fuzz_base, _load_both

    R, hb, hs, bg = model
    O = oracle_mod
    n = 128
    base, pv = _fuzz_base(O, gametype, n, np.random.default_rng(seed))
    # a shell's heading is what its velocity says (set once, at launch: SRC/game.cpp:263); the device keeps the velocity only
    ang = np.degrees(np.arctan2(base["shell_vy"], base["shell_vx"]))
    base["shell_angle"] = np.where(ang < 0, ang + 360.0, ang)
    env, orc = _load_both(sfa, O, gametype, base, prev_vlner=pv)
    raw = env.render("image-raw").cpu().numpy()
    small = env.render("image").cpu().numpy()
    snaps = orc.snapshots()
    assert (snaps["missile_alive"][:, 19] == 1).any() and (snaps["shell_alive"][:, 16:] == 1).any()
    assert (snaps["missile_alive"].sum(1) >= 15).any() and (snaps["shell_alive"].sum(1) >= 15).any()
    for i in range(n):
        want = R.render_raw(snaps[i], hb, hs, bg=bg)
        frames_close(raw[i], want, ("raw", gametype, i))
        frames_close(small[i, 0], R.resize_area(want), ("84x84", gametype, i))
    env.close()
    monkeypatch.setenv("SFMI_NO_EXPLOSION_CACHE", "1")
    monkeypatch.setenv("SFMI_NO_RENDER_ORDER", "1")
    plain, _ = _load_both(sfa, O, gametype, base, prev_vlner=pv)
    assert np.array_equal(plain.render("image-raw").cpu().numpy(), raw)
    assert np.array_equal(plain.render("image").cpu().numpy(), small)
    plain.close()


def _live_record_bytes(rec):
    """The bytes of draw records [N, 432] (sf_drawrec.h) that mean something: the header, the ship's position, the positions
    and headings of the LIVE missile slots (a dead slot's entry keeps whatever was there; the fortress's piece is unused)."""
    n = rec.shape[0]
    assert rec.shape[1] == 432
    live = np.zeros((n, 432), bool)
    live[:, :48] = True  # the header's two pieces and the ship's position
    objmask = rec[:, 12:16].copy().view(np.uint32)[:, 0]
    for s in range(20):
        on = ((objmask >> (2 + s)) & 1).astype(bool)[:, None]
        live[:, 64 + 16 * s:80 + 16 * s] = on
        live[:, 384 + 2 * s:386 + 2 * s] = on
    return np.where(live, rec, 0)


@pytest.mark.parametrize("gametype,policy", [("youturn", "random"), ("autoturn", "hunter"), ("youturn", "charger")])
def test_draw_records_of_the_step_equal_records_from_the_state(sfa, gametype, policy):
    """The frame kernel reads the envs' DRAW RECORDS (sf_drawrec.h), which an image batch's step launches leave behind
    (sf_step_kernel, with the state in registers) and which a frame rebuilds from the state in HBM after any other change
    (sf_drawrec_kernel).  Two pieces of code, one set of functions: after every few steps of three kinds of play both ways
    give the same bytes (header, ship, fortress, live missiles) -- and the same frames."""
    from sfscript import open_loop_actions

    N, T = 1024, 900
    rng = np.random.default_rng(5)
    env = sfa.SFVecEnv(N, gametype=gametype, obs_type="image", spawn_stride=1)
    env.reset()
    acts = torch.from_numpy(open_loop_actions(policy, (T, N), env.n_actions, rng, phase=rng.integers(0, 96, N))).to(env.device)
    seen_flags = 0
    for t in range(T):
        obs, *_ = env.step_tensors(acts[t])
        if t % 37 == 0 or t > T - 30:
            a = env.draw_records(False)
            frame_a = obs.clone()
            b = env.draw_records(True)
            assert np.array_equal(_live_record_bytes(a), _live_record_bytes(b)), (t, np.argwhere(_live_record_bytes(a) != _live_record_bytes(b))[:8])
            assert torch.equal(env.render("image"), frame_a), t
            seen_flags |= int(np.bitwise_or.reduce(a[:, 20:24].copy().view(np.uint32)[:, 0]))
    # the play reached the decisions the records exist for: dead ships, a picture-less fortress, score / bar off the baked path
    assert seen_flags & (1 << 12) and seen_flags & (1 << 13)
    if policy == "hunter":
        assert seen_flags & (3 << 20), hex(seen_flags)  # a destroyed fortress's explosion
    env.close()


def test_image_batches_of_every_workgroup_size(sfa):
    """An image batch's step launches go out as 64-, 128- or 256-thread workgroups by the batch's size (sf_launch_step):
    three instantiations writing draw records.  Lane i plays spawn i in every batch (spawn_stride 1), so the first 4 096
    envs of batches of 4 096, 20 480 and 36 864 show the same frames and leave the same records, and in the largest the
    records of the step equal the records rebuilt from the state."""
    from sfscript import open_loop_actions

    n0, T = 4096, 260
    rng = np.random.default_rng(12)
    acts0 = open_loop_actions("hunter", (T, n0), 3, rng, phase=rng.integers(0, 96, n0))
    ref = {}
    for n in (n0, 20480, 36864):
        env = sfa.SFVecEnv(n, gametype="autoturn", obs_type="image", spawn_stride=1)
        env.reset()
        acts = np.concatenate([acts0, open_loop_actions("hunter", (T, n - n0), 3, rng, phase=rng.integers(0, 96, n - n0))], axis=1) if n > n0 else acts0
        a = torch.from_numpy(np.ascontiguousarray(acts)).to(env.device)
        for t in range(T):
            obs, *_ = env.step_tensors(a[t])
            if t % 29 == 0 or t == T - 1:
                rec = _live_record_bytes(env.draw_records(False)[:n0])
                frame = obs[:n0].cpu().numpy()
                if n == n0:
                    ref[t] = (rec, frame)
                else:
                    assert np.array_equal(rec, ref[t][0]), (n, t, np.argwhere(rec != ref[t][0])[:6].tolist())
                    assert np.array_equal(frame, ref[t][1]), (n, t)
        if n != n0:
            assert np.array_equal(_live_record_bytes(env.draw_records(False)), _live_record_bytes(env.draw_records(True))), n
        env.close()
    assert any(r[1].max() > 0 for r in ref.values())


@pytest.mark.parametrize("scale,viewport,ls", [(.25, (100, 60, 500, 520), 2), (.3, (130, 80, 450, 460), 4.5)])
def test_frames_in_another_geometry_vs_model(sfa, oracle_mod, scale, viewport, ls):
    """SSF_Env(scale, viewport, ls) (ENV:50-60): any geometry but the default one goes through the general renderer
    (sf_render_generic.hip).  Frames of both sizes -- the raw [h][w] surface and the trainer's 84x84 INTER_AREA image --
    against the numpy model parametrised the same way, on oracle lock-step states with everything on screen (explosions,
    missiles, shells, a score, a filling bar); and the SSF_Env surface: shapes follow the geometry, the default still
    takes the fast kernel, a surface the trainer's resize cannot shrink is refused."""
    from oracle import render_np as R
    from sfscript import open_loop_actions

    O = oracle_mod
    z = np.load(os.path.join(GOLDEN, "tables.npz"))
    hb, hs = z["hex_points"][:12], z["hex_points"][12:]
    w, h = int(viewport[2] * scale), int(viewport[3] * scale)
    N, T = 6, 420
    rng = np.random.default_rng(8)
    env = sfa.SFVecEnv(N, gametype="autoturn", obs_type="image-raw", spawn_stride=1, image_geometry=(scale, viewport, ls))
    assert env.obs_shape == (h, w) and env.observation_space.shape == (h, w)
    orc = O.OracleVecEnv("autoturn", N, spawn_stride=1)
    acts = open_loop_actions("hunter", (T, N), env.n_actions, rng, phase=rng.integers(0, 96, N))
    prev = R.set_geometry(scale, viewport, ls)
    try:
        A = R.glyphs_for_geometry()  # both geometries are among score_glyphs.npz's: the text is the reference's here too
        env.set_score_glyphs(A["alpha"], A["layout"], A["x0"])
        bg = R.background(hb, hs)
        checked = 0
        for t in range(T):
            raw, *_ = env.step_tensors(torch.from_numpy(acts[t]).to(env.device))
            orc.step(acts[t].astype(np.int32))
            if t % 30 == 29 or t > T - 4:
                small = env.render("image").cpu().numpy()
                raw = raw.cpu().numpy()
                snaps = orc.snapshots()
                for lane in range(0, N, 2):
                    want = R.render_raw(snaps[lane], hb, hs, bg=bg)
                    frames_close(raw[lane], want, ("raw", t, lane))
                    frames_close(small[lane, 0], R.resize_area(want), ("84x84", t, lane))
                    assert np.array_equal(small[lane, 0], R.resize_area(raw[lane]))  # the device's resize of the device's frame: exact
                    checked += 1
        assert checked >= 40
    finally:
        R.set_geometry(*prev)
    # back to the default geometry on the same batch: the fast kernel's 92 x 90 frames again
    env.set_image_geometry()
    assert env.obs_shape == (92, 90) and env.render("image-raw").shape == (N, 92, 90)
    env.close()
    e1 = sfa.SSF_Env("youturn", scale=scale, viewport=viewport, ls=ls, obs_type="image")
    o = e1.reset()
    assert o.shape == (h, w) and e1.observation_space.shape == (h, w, 3) and e1.render("rgb_array").shape == (h, w, 3)
    o2, r, d, i = e1.step(1)
    assert o2.shape == (h, w) and o2.max() == 255 or o2.max() > 100
    e1.close()
    with pytest.raises(ValueError):
        sfa.SSF_Env("youturn", scale=.1, obs_type="image")  # a 45 x 46 surface: nothing to shrink to 84 x 84
    with pytest.raises(ValueError):  # more than 0.75 pixels per unit: cairo's circle takes more Bezier segments than the renderer's
        sfa.SSF_Env("youturn", scale=1.0, viewport=(255, 215, 200, 200), obs_type="image")
    e2 = sfa.SSF_Env("youturn", scale=.1, obs_type="features")  # ... but fine where no frame is drawn
    with pytest.raises(ValueError):
        e2.render("rgb_array")
    e2.close()


def test_frames_after_every_way_of_stepping(sfa):
    """The draw records the frame kernel reads are left by the step launch that produced the state -- whichever entry point that
    was: one step, a fused rollout (the LAST tick's records), sampled actions, the trainer's recorded step, a state edited
    through set_field, a reset.  After each, the frame of an image batch equals the frame of a twin batch that was stepped
    one tick at a time with the same actions, and the records equal records rebuilt from the state."""
    N = 320
    rng = np.random.default_rng(12)
    a = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=1)
    b = sfa.SFVecEnv(N, gametype="youturn", obs_type="image", spawn_stride=1)
    acts = torch.from_numpy(rng.integers(0, 5, (400, N)).astype(np.uint8)).to(a.device)

    def same(what):
        fa, fb = a.render("image"), b.render("image")
        assert torch.equal(fa, fb), what
        ra = a.draw_records(False)
        assert np.array_equal(_live_record_bytes(ra), _live_record_bytes(a.draw_records(True))), what
        assert torch.equal(a.render("image-raw"), b.render("image-raw")), what

    t = 0
    for _ in range(150):  # into mid-game states
        a.step_tensors(acts[t]); b.step_tensors(acts[t]); t += 1
    same("single steps")
    a.rollout(acts[t:t + 37], want_obs=False)  # fused: 37 ticks in one launch
    for k in range(37):
        b.step_tensors(acts[t + k])
    t += 37
    same("fused rollout")
    a.seed_actions(5); b.seed_actions(5)
    played = torch.empty(N, dtype=torch.uint8, device=a.device)
    for _ in range(9):
        a.step_sampled(actions_out=played)
        b.step_sampled()
    same("sampled steps")
    _, _, _, _, pa = a.rollout_sampled(11, want_obs=False)
    for k in range(11):
        b.step_tensors(pa[k])
    b.seed_actions(5)  # (keep the two samplers in step: b did not draw those 11 ticks)
    a.seed_actions(5)
    same("fused sampled rollout")
    # an edited state: the records are rebuilt from the state by the next frame
    for e in (a, b):
        x = e.get_field("ship_x")
        x[::7] += 3.25
        e.set_field("ship_x", x)
    same("set_field")
    a.step_tensors(acts[t]); b.step_tensors(acts[t]); t += 1
    same("a step after set_field")
    ra, rb = a.reset(), b.reset()
    assert torch.equal(ra, rb)
    same("reset")
    a.close()
    b.close()


@pytest.mark.parametrize("gametype", ["autoturn", "youturn"])
def test_an_image_batch_plays_the_same_game_as_a_features_batch(sfa, gametype):
    """The step launches of an image batch run another instantiation of the step kernel than a features batch -- no
    observation epilogue, but the draw records for the frame kernel, with projectile-near-the-HUD flags riding on the missile
    pool's event words -- and (16 384 envs and fewer) 64-thread workgroups.  Whatever they add must not touch the game: after
    hundreds of steps of fortress-killing play the two batches hold the same state bit for bit and have paid out the same
    rewards."""
    from sfscript import open_loop_actions

    N, T = 4096, 700
    rng = np.random.default_rng(31)
    img = sfa.SFVecEnv(N, gametype=gametype, obs_type="image", spawn_stride=1)
    fea = sfa.SFVecEnv(N, gametype=gametype, obs_type="features", spawn_stride=1)
    acts = torch.from_numpy(open_loop_actions("hunter", (T, N), img.n_actions, rng, phase=rng.integers(0, 96, N))).to(img.device)
    ri = torch.zeros(N, dtype=torch.int64, device=img.device)
    rf = torch.zeros_like(ri)
    ki = torch.zeros_like(ri)
    for t in range(T):
        _, r1, d1, i1 = img.step_tensors(acts[t])
        _, r2, d2, i2 = fea.step_tensors(acts[t])
        ri += r1
        rf += r2
        ki += i1
        if t % 97 == 0:
            assert torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(i1, i2), t
    assert torch.equal(ri, rf) and int(ki.sum()) > (200 if gametype == "autoturn" else 0)
    a, b = img.state_dict(), fea.state_dict()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    img.close()
    fea.close()


@pytest.mark.gpu
def test_shells_on_exact_degree_headings(sfa, model):
    """the frame kernels re-derive a shell's drawn heading from its velocity, (int)(atan2(vy, vx) * 180 / pi), where the
    reference truncates the heading it stored when the shell was fired (SRC/game.cpp:159-173, SRC/draw.cpp:248-252).  A heading
    that is a whole number of degrees must not land an ulp below it: shells flying along the axes and the diagonals -- the only
    whole-degree bearings a fortress at whole coordinates has to a ship at whole coordinates (a game's first tick) -- with the
    velocities the step kernel gives them, 6 (d / |d|), in both frame kernels."""
    R, hb, hs, bg = model
    from oracle import oracle as O
    dirs = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]
    base = np.load(os.path.join(GOLDEN, "frames", "scores.npz"))["base"]
    snaps = np.repeat(base.reshape(1), len(dirs))
    for i, (dx, dy) in enumerate(dirs):
        ddx, ddy = 60.0 * dx, 60.0 * dy
        nrm = np.sqrt(ddx * ddx + ddy * ddy)
        snaps["shell_alive"][i][2] = 1
        snaps["shell_x"][i][2], snaps["shell_y"][i][2] = 355.0 + ddx, 315.0 + ddy
        snaps["shell_vx"][i][2], snaps["shell_vy"][i][2] = 6.0 * (ddx / nrm), 6.0 * (ddy / nrm)
        snaps["shell_angle"][i][2] = 45.0 * i  # what the reference stored: rad2deg(atan2(dy, dx)), exact for these
    for geom in (None, (.25, (100, 60, 500, 520), 2)):
        env = sfa.SFVecEnv(len(dirs), gametype="youturn", obs_type="image-raw", image_geometry=geom)
        _load_snaps(env, snaps)
        got = env.render("image-raw").cpu().numpy()
        prev = R.set_geometry(*geom) if geom else None
        try:
            for i in range(len(dirs)):
                frames_close(got[i], R.render_raw(snaps[i], hb, hs, text=True if geom is None else "segments"), ("heading", 45 * i, geom))
        finally:
            if prev:
                R.set_geometry(*prev)
        env.close()


@pytest.mark.gpu
def test_explosion_prepass_equals_drawing_in_place():
    """sf_explosion_kernel (batches up to 6 144 envs: a freshly dead ship's explosion drawn by eight waves into its cache entry
    ahead of the frame kernel) against the frame kernel drawing it itself: the same run -- 512 envs, 400 random steps, some two
    thousand deaths -- with SFMI_EXPLOSION_PREPASS=1 and =0 (read once per process: two child processes), every frame's
    fingerprint equal (tools/render_hash.py: a changed byte anywhere changes a line)."""
    import subprocess
    import sys

    from conftest import ROOT

    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, SFMI_EXPLOSION_PREPASS=flag)
        r = subprocess.run([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from render_hash import fingerprints; "
                            "print('\\n'.join(fingerprints('youturn', 512, 400, 'random')))" % (ROOT, os.path.join(ROOT, "tools"))],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l and l[0].isdigit()])
    assert len(outs[0]) == len(outs[1]) == 400
    bad = [i for i, (a, b) in enumerate(zip(*outs)) if a != b]
    assert not bad, "frames with and without the pre-pass differ at steps %s ..." % bad[:5]
