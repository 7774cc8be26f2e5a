"""Per-step fingerprints of the image path, to compare two BUILDS of libsfmi (render_soak.py compares two configurations of
one build): same seeds, same actions -> the same states, so every frame must hash the same.
    SFMI_LIB_PATH=build/abl/libsfmi_a.so python tools/render_hash.py OUT_A.txt [gametype] [lanes] [steps] [random|hunter]
    SFMI_LIB_PATH=build/abl/libsfmi_b.so python tools/render_hash.py OUT_B.txt ...;  cmp OUT_A.txt OUT_B.txt
One line per step: a 63-bit weighted sum over every byte of the [lanes, 84, 84] frame (weights from a fixed seed: a single
changed byte changes the sum), every 8th step also of the raw 90x92 frames."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def fingerprints(gametype="youturn", N=4096, T=3000, policy="hunter", text="atlas"):
    """One line per step, see the module docstring.  text: "atlas" = the built-in glyph atlas (the reference's text),
    "segments" = the seven-segment fallback, by name (what every build before round 6 drew)."""
    from spacefortress_amd import SFVecEnv
    env = SFVecEnv(N, gametype=gametype, obs_type="image", spawn_stride=3)
    if text == "segments":
        env.set_score_glyphs(None)
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(11)
    w84 = torch.randint(1, 1 << 31, (N, 84 * 84), generator=g, dtype=torch.int64).to(env.device)
    wraw = None
    rng = np.random.default_rng(5)
    phase = rng.integers(0, 96, N)
    pat = np.array(([1] + [0] * 7) * 11 + [1, 0, 1, 0] + [0] * 4, np.uint8)
    lines = []
    for t in range(T):
        acts = rng.integers(0, env.n_actions, N).astype(np.uint8)
        if policy == "hunter":
            acts = np.where(rng.random(N) < 0.1, acts, pat[(t + phase) % len(pat)]).astype(np.uint8)
        o, *_ = env.step_tensors(torch.from_numpy(acts).to(env.device))
        f = o.reshape(N, -1)
        h = int((f.to(torch.int64) * w84).sum().item()) & ((1 << 63) - 1)
        line = "%d %x" % (t, h)
        if t % 8 == 0:
            r = env.render("image-raw").reshape(N, -1)
            if wraw is None:
                wraw = torch.randint(1, 1 << 31, r.shape, generator=g, dtype=torch.int64).to(env.device)
            line += " %x" % (int((r.to(torch.int64) * wraw).sum().item()) & ((1 << 63) - 1))
        lines.append(line)
    env.close()
    return lines


if __name__ == "__main__":
    out = sys.argv[1]
    gametype = sys.argv[2] if len(sys.argv) > 2 else "youturn"
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
    policy = sys.argv[5] if len(sys.argv) > 5 else "hunter"
    text = sys.argv[6] if len(sys.argv) > 6 else "atlas"
    lines = fingerprints(gametype, N, T, policy, text)
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote %d step fingerprints of %d lanes (%s, %s) to %s" % (T, N, gametype, policy, out))
