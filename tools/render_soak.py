"""Long equality run of the render kernel's shortcuts: a batch with every cache, picture and launch-order hint (the
product) against a batch that draws everything in place in env order (SFMI_NO_EXPLOSION_CACHE, SFMI_NO_RENDER_ORDER), both
stepped with the same actions; every frame of every env must be identical, in both sizes.
    python tools/render_soak.py [gametype] [lanes] [steps] [random|hunter]
`hunter` (see tools/soak.py) destroys the fortress thousands of times: scores other than 0, every bar state, the fortress's
explosion picture."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gametype = sys.argv[1] if len(sys.argv) > 1 else "autoturn"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
policy = sys.argv[4] if len(sys.argv) > 4 else "hunter"
os.environ["SFMI_NO_EXPLOSION_CACHE"] = "1"
os.environ["SFMI_NO_RENDER_ORDER"] = "1"
from spacefortress_amd import SFVecEnv
plain = SFVecEnv(N, gametype=gametype, obs_type="image", spawn_stride=3)
plain.render("image")  # the switches are read at create / by the first frame
del os.environ["SFMI_NO_EXPLOSION_CACHE"], os.environ["SFMI_NO_RENDER_ORDER"]
prod = SFVecEnv(N, gametype=gametype, obs_type="image", spawn_stride=3)
rng = np.random.default_rng(5)
phase = rng.integers(0, 96, N)
pat = np.array(([1] + [0] * 7) * 11 + [1, 0, 1, 0] + [0] * 4, np.uint8)
t0 = time.time(); seen_scores = set(); seen_bar = set(); fort_dead = 0
for t in range(T):
    acts = rng.integers(0, prod.n_actions, N).astype(np.uint8)
    if policy == "hunter":
        acts = np.where(rng.random(N) < 0.1, acts, pat[(t + phase) % len(pat)]).astype(np.uint8)
    a = torch.from_numpy(acts).to(prod.device)
    o1, *_ = plain.step_tensors(a)
    o2, *_ = prod.step_tensors(a)
    if not torch.equal(o1, o2):
        bad = (o1 != o2).flatten(1).any(1).nonzero().flatten()[:8].tolist()
        raise SystemExit("84x84 frames differ at step %d, envs %s" % (t, bad))
    if t % 8 == 0 and not torch.equal(plain.render("image-raw"), prod.render("image-raw")):
        raise SystemExit("raw frames differ at step %d" % t)
    if t % 50 == 0:
        seen_scores.update(np.unique(prod.get_field("points").astype(np.int64)).tolist())
        seen_bar.update(np.unique(np.minimum(prod.get_field("vlner"), 11)).tolist())
        fort_dead += int(((prod.get_field("flags").astype(np.int64) & 2) == 0).sum())
    if t % 500 == 499:
        print("step %d ok (%.0f s)" % (t + 1, time.time() - t0), flush=True)
print("RENDER SOAK OK: %s (%s), %d lanes x %d steps = %.1fM frames in both sizes; scores seen %d..%d (%d values), bar states %s, "
      "fortress-dead samples %d" % (gametype, policy, N, T, N * T / 1e6, min(seen_scores), max(seen_scores), len(seen_scores),
                                     sorted(int(x) for x in seen_bar), fort_dead))
