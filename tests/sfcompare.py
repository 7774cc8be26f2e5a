"""Helpers shared by the GPU parity tests: HIP batch state <-> oracle snapshots."""
import numpy as np

FL_SHIP, FL_FORT, FL_FIRE, FL_THRUST, FL_LEFT, FL_RIGHT = 1, 2, 4, 8, 16, 32

INT_FIELDS = [  # (hip field, snapshot field)
    ("time", "time"), ("ship_death_timer", "ship_death_timer"), ("fire_timer", "fire_timer"),
    ("thrust_timer", "thrust_timer"), ("left_timer", "left_timer"), ("right_timer", "right_timer"),
    ("fort_timer", "fort_timer"), ("fort_death_timer", "fort_death_timer"),
    ("fort_vuln_timer", "fort_vuln_timer"), ("vlner", "vlner"),
]
EXACT_F_FIELDS = [("ship_x", "ship_x"), ("ship_y", "ship_y"), ("ship_vx", "ship_vx"), ("ship_vy", "ship_vy"),
                  ("points", "points"), ("raw_points", "raw_points")]


def obs_close(a, b, f64):
    """Device observation vs the oracle's float64 one: 1e-9 relative for float64 batches, float32 rounding otherwise."""
    tol = 1e-9 if f64 else 1e-5
    scale = np.maximum(1.0, np.abs(b))
    return np.abs(a.astype(np.float64) - b) <= tol * scale + (0 if f64 else 4e-5)


def mask_bits(mask, n=20):
    return ((mask[None, :] >> np.arange(n, dtype=np.uint32)[:, None]) & 1).astype(bool)  # [slot, env]


def compare_state(sd, snaps, shell_tol=1e-9, lanes=None):
    """sd: SFVecEnv.state_dict(); snaps: oracle SNAPSHOT_DTYPE[n].  Returns a list of mismatches.
    Bit-exact for everything except shell positions/velocities (real device sin/cos feeds them)."""
    bad = []
    if lanes is not None:
        sd = {k: np.asarray(v)[..., lanes] for k, v in sd.items()}
    sel = slice(None)

    def chk(name, a, b, exact=True, tol=0.0):
        a = np.asarray(a)
        if exact:
            ok = a.tobytes() == np.ascontiguousarray(b, a.dtype).tobytes() if a.dtype.kind == "f" else np.array_equal(a, b)
            if not ok:
                idx = np.flatnonzero(np.atleast_1d((a != b).reshape(-1)))
                bad.append((name, int(idx.size), idx[:5].tolist()))
        else:
            d = np.abs(a - b)
            if d.size and d.max() > tol:
                bad.append((name, float(d.max())))

    for h, s in INT_FIELDS:
        chk(h, sd[h], snaps[s])
    for h, s in EXACT_F_FIELDS:
        chk(h, sd[h], snaps[s])
    fl = sd["flags"][sel]
    chk("ship_alive", (fl & FL_SHIP) != 0, snaps["ship_alive"] != 0)
    chk("fort_alive", (fl & FL_FORT) != 0, snaps["fort_alive"] != 0)
    chk("fire_flag", (fl & FL_FIRE) != 0, snaps["fire_flag"] != 0)
    chk("thrust_flag", (fl & FL_THRUST) != 0, snaps["thrust_flag"] != 0)
    chk("left_flag", (fl & FL_LEFT) != 0, snaps["left_flag"] != 0)
    chk("right_flag", (fl & FL_RIGHT) != 0, snaps["right_flag"] != 0)
    chk("ship_angle", sd["ship_angle"][sel].astype(np.float64), snaps["ship_angle"])
    chk("fort_angle", sd["fort_angle"][sel].astype(np.float64), snaps["fort_angle"])
    chk("fort_last_angle", sd["fort_last_angle"][sel].astype(np.float64), snaps["fort_last_angle"])
    chk("stats", sd["stats"][:, sel].T, snaps["stats"])
    ml = mask_bits(sd["missile_mask"][sel])
    sl = mask_bits(sd["shell_mask"][sel])
    chk("missile_alive", ml.T, snaps["missile_alive"] != 0)
    chk("shell_alive", sl.T, snaps["shell_alive"] != 0)
    oml = (snaps["missile_alive"] != 0).T
    osl = (snaps["shell_alive"] != 0).T
    if np.array_equal(ml, oml):
        for h, s in (("missile_x", "missile_x"), ("missile_y", "missile_y")):
            a = sd[h][:, sel][ml]
            b = snaps[s].T[oml]
            if a.tobytes() != b.tobytes():
                bad.append((h, int((a != b).sum())))
        a = sd["missile_angle"][:, sel][ml].astype(np.float64)
        if not np.array_equal(a, snaps["missile_angle"].T[oml]):
            bad.append(("missile_angle",))
    if np.array_equal(sl, osl):
        for h in ("shell_x", "shell_y", "shell_vx", "shell_vy"):
            a = sd[h][:, sel][sl]
            b = snaps[h].T[osl]
            if a.size and np.abs(a - b).max() > shell_tol:
                bad.append((h, float(np.abs(a - b).max())))
    return bad


def snapshots_to_fields(snaps, tick_ms=34):
    """Oracle snapshots -> dict of HIP field arrays (for SFVecEnv.set_field)."""
    n = len(snaps)
    f = {}
    for h, s in INT_FIELDS:
        f[h] = snaps[s].astype(np.int32)
    for h, s in EXACT_F_FIELDS:
        f[h] = snaps[s].copy()
    fl = np.zeros(n, np.uint8)
    for bit, s in ((FL_SHIP, "ship_alive"), (FL_FORT, "fort_alive"), (FL_FIRE, "fire_flag"),
                   (FL_THRUST, "thrust_flag"), (FL_LEFT, "left_flag"), (FL_RIGHT, "right_flag")):
        fl |= np.where(snaps[s] != 0, bit, 0).astype(np.uint8)
    f["flags"] = fl
    f["ship_angle"] = snaps["ship_angle"].astype(np.int16)
    f["fort_angle"] = snaps["fort_angle"].astype(np.int16)
    f["fort_last_angle"] = snaps["fort_last_angle"].astype(np.int16)
    f["stats"] = np.ascontiguousarray(snaps["stats"].T.astype(np.int32))
    w = (1 << np.arange(20, dtype=np.uint32))
    f["missile_mask"] = ((snaps["missile_alive"] != 0) * w[None, :]).sum(1).astype(np.uint32)
    f["shell_mask"] = ((snaps["shell_alive"] != 0) * w[None, :]).sum(1).astype(np.uint32)
    for h in ("missile_x", "missile_y", "shell_x", "shell_y", "shell_vx", "shell_vy"):
        f[h] = np.ascontiguousarray(snaps[h].T)
    f["missile_angle"] = np.ascontiguousarray(snaps["missile_angle"].T.astype(np.int16))
    return f
