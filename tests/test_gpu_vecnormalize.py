"""SFVecNormalize (sf_normalize.hip) against the numpy restatement of gym_vecenv.VecNormalize
(oracle/vecnorm_np.py; the package itself is not in the reference tree: parity unpinned beyond that).
Tolerance: 1e-6 absolute on normalised float32 observations / rewards (float64 math inside, one-pass
batch variance vs numpy's two-pass), 1e-9 relative on the running statistics."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def sfa():
    import spacefortress_amd as m
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return m


@pytest.mark.parametrize("gametype,obs_type,f64,n", [("youturn", "features", False, 4096), ("autoturn", "features", True, 1000),
                                                    ("youturn", "monitors", False, 333)])
def test_vecnormalize_matches_numpy_model(sfa, gametype, obs_type, f64, n):
    from oracle import vecnorm_np as V
    rng = np.random.default_rng(4)
    mk = lambda: sfa.SFVecEnv(n, gametype=gametype, obs_type=obs_type, spawn_stride=1,
                              obs_dtype=torch.float64 if f64 else torch.float32)
    raw, env = mk(), sfa.SFVecNormalize(mk())
    model = V.VecNormalize(n, (raw.obs_dim,))
    o_raw = raw.reset().cpu().numpy().astype(np.float64)
    o = env.reset().cpu().numpy()
    assert np.abs(o - model.reset(o_raw)).max() < 1e-6
    for t in range(300):
        a = torch.from_numpy(rng.integers(0, raw.n_actions, n).astype(np.uint8)).to(raw.device)
        o_raw, r_raw, d_raw, i_raw = raw.step_tensors(a)
        o, r, d, i = env.step_tensors(a)
        mo, mr = model.step(o_raw.cpu().numpy().astype(np.float64), r_raw.cpu().numpy().astype(np.float64))
        assert torch.equal(d, d_raw) and torch.equal(i, i_raw)
        assert np.abs(o.cpu().numpy() - mo).max() < 1e-6, t
        assert np.abs(r.cpu().numpy() - mr).max() < 1e-6, t
    ob, rt = env.ob_rms, env.ret_rms
    assert np.allclose(ob.mean, model.ob_rms.mean, rtol=1e-9, atol=1e-12) and np.allclose(ob.var, model.ob_rms.var, rtol=1e-9, atol=1e-12)
    assert ob.count == model.ob_rms.count and rt.count == model.ret_rms.count
    assert np.isclose(rt.mean, model.ret_rms.mean, rtol=1e-9) and np.isclose(rt.var, model.ret_rms.var, rtol=1e-9)
    assert np.allclose(env.ret, model.ret, rtol=1e-12)
    # frozen statistics (evaluation) and a state round trip
    sd = env.state_dict()
    env.training = False
    a = torch.zeros(n, dtype=torch.uint8, device=raw.device)
    o_raw, r_raw, _, _ = raw.step_tensors(a)
    o, r, _, _ = env.step_tensors(a)
    want = np.clip((o_raw.cpu().numpy() - model.ob_rms.mean) / np.sqrt(model.ob_rms.var + 1e-8), -10, 10)
    assert np.abs(o.cpu().numpy() - want).max() < 1e-6
    assert np.array_equal(env.state_dict()["stats"], sd["stats"])
    env.load_state_dict(sd)
    assert np.array_equal(env.state_dict()["ret"], sd["ret"])
    # the numpy-in / numpy-out surface of the reference's wrapper
    env.training = True
    on, rn, dn, inn = env.step(np.zeros(n, np.int64))
    assert on.shape == (n, raw.obs_dim) and rn.dtype == np.float64 and dn.dtype == bool
    env.close()
    raw.close()


def test_vecnormalize_rejects_images(sfa):
    v = sfa.SFVecEnv(4, obs_type="image")
    with pytest.raises(ValueError):
        sfa.SFVecNormalize(v)
    v.close()
