import sys, os, math
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/spacefortress_amd") else os.environ.get("GRAFT_REPO_ROOT","."))
from spacefortress_amd import SFVecEnv
n = 16384
env = SFVecEnv(n, gametype="youturn", obs_type="features", spawn_stride=1)
env.reset()
acts = torch.randint(0, 5, (64, n), device=env.device, dtype=torch.uint8)
for t in range(400): env.step_tensors(acts[t % 64])
flags = env.get_field("flags").astype(np.int64)
x = env.get_field("ship_x"); y = env.get_field("ship_y"); fa = env.get_field("fort_angle").astype(np.int64)
alive = (flags & 1) != 0; falive = (flags & 2) != 0
S = 0.2; VX, VY = 130, 80
gx = (x - VX) * S; gy = (y - VY) * S
def box_live(gx, gy):
    ext = 27 * S
    return np.floor(gx - ext), np.floor(gy - ext), np.ceil(gx + ext), np.ceil(gy + ext)
def box_expl(gx, gy):
    ext = 64.5 * S
    return np.maximum(np.floor(gx - ext), 0), np.maximum(np.floor(gy - ext), 0), np.minimum(np.ceil(gx + ext), 90), np.minimum(np.ceil(gy + ext), 92)
bx0 = np.where(alive, box_live(gx, gy)[0], box_expl(gx, gy)[0]); by0 = np.where(alive, box_live(gx, gy)[1], box_expl(gx, gy)[1])
bx1 = np.where(alive, box_live(gx, gy)[2], box_expl(gx, gy)[2]); by1 = np.where(alive, box_live(gx, gy)[3], box_expl(gx, gy)[3])
def meets(a0, b0, a1, b1):
    return (bx0 < a1) & (a0 < bx1) & (by0 < b1) & (b0 < by1)
full = meets(37 - 3, 39 - 3, 53 + 3, 55 + 3)
# tight boxes per sector
lines = [(0, 0, 36, 0), (0, -18, 18, -18), (18, -18, 18, 18), (18, 18, 0, 18)]
tb = np.zeros((36, 4))
for s in range(36):
    a = math.radians(10 * s); ca, sa = math.cos(a), math.sin(a)
    xs, ys = [], []
    for (ax, ay, bx, by) in lines:
        ux, uy = bx - ax, by - ay; L = math.hypot(ux, uy); nx, ny = -uy / L * 1.5, ux / L * 1.5
        for (px, py) in ((ax + nx, ay + ny), (bx + nx, by + ny), (bx - nx, by - ny), (ax - nx, ay - ny)):
            xs.append((355 + ca * px - sa * py - VX) * S); ys.append((315 + sa * px + ca * py - VY) * S)
    tb[s] = (math.floor(min(xs)), math.floor(min(ys)), math.ceil(max(xs)), math.ceil(max(ys)))
sec = np.clip(fa // 10, 0, 35)
t = tb[sec]
tight = meets(t[:, 0] - 3, t[:, 1] - 3, t[:, 2] + 3, t[:, 3] + 3)
print("ship alive %.3f fortress alive %.3f" % (alive.mean(), falive.mean()))
print("in place (full box):  all %.3f | live ship %.3f | dead ship %.3f" % ((full & falive).mean(), (full & falive & alive).mean(), (full & falive & ~alive).mean()))
print("in place (tight box): all %.3f | live ship %.3f | dead ship %.3f" % ((tight & falive).mean(), (tight & falive & alive).mean(), (tight & falive & ~alive).mean()))
print("tight box sizes:", (tb[:, 2] - tb[:, 0]).mean(), (tb[:, 3] - tb[:, 1]).mean())
