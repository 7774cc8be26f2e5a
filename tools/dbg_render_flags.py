import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from spacefortress_amd import SFVecEnv
n=16384
env = SFVecEnv(n, gametype="youturn", obs_type="image", spawn_stride=1, reuse_buffers=True)
env.reset()
acts = torch.randint(0, env.n_actions, (64, n), device=env.device, dtype=torch.uint8)
for t in range(400): o,_,_,_ = env.step_tensors(acts[t % 64])
raw = o.reshape(n, -1)[:, :72].cpu().numpy()
f = raw[:, :2]
T = raw[:, 4:20].copy().view(np.uint32).astype(np.float64)  # wave life, to barrier, to after strokes, to after shells
names0 = ["baked_text","baked_bar","near_text","near_bar","close_text","close_bar","ship_alive","explosion_done"]
names1 = ["variant_restart","fort_pic","fort_alive","smask","mmask","pnts!=0","vlner!=0","xst.flags"]
for j,names in enumerate((names0,names1)):
    for b,nm in enumerate(names): print("%-16s %.3f" % (nm, ((f[:,j]>>b)&1).mean()))

print("wave clocks: total %.0f  | to barrier %.0f  strokes done %.0f  shells done %.0f" % tuple(T.mean(0)))
cls = {"ship alive": (f[:,0]>>6)&1, "dead": 1-((f[:,0]>>6)&1), "smask": (f[:,1]>>3)&1, "no smask": 1-((f[:,1]>>3)&1), "baked both": ((f[:,0]&3)==3), "bar not baked": ((f[:,0]>>1)&1)==0, "text not baked": (f[:,0]&1)==0}
for k,m in cls.items():
    m = m.astype(bool)
    print("%-16s n=%5d  total %.0f  barrier %.0f strokes %.0f shells %.0f  hud+end %.0f" % ((k, m.sum()) + tuple(T[m].mean(0)) + ((T[m,0]-T[m,3]).mean(),)))

P = raw[:, 20:36].copy().view(np.uint32).astype(np.float64)
print("prologue stamps (clocks from wave start): state decoded %.0f | mtab+shells done, before DMA %.0f | strokes built+tests %.0f | before final wait %.0f | barrier passed %.0f" % (tuple(P.mean(0)) + (T[:,1].mean(),)))

D = raw[:, 36:56].copy().view(np.uint32).astype(np.float64)
ok = (D < 1e6).all(1) & (D[:, 4] > 0)
nch = raw[:, 2]
for c in sorted(set(nch[ok].tolist())):
    m = ok & (nch == c)
    d = D[m].mean(0)
    print("draw_strokes, main call, %d chunk(s): n=%5d  entry %.0f | writing records %.0f | cheap rounds %.0f | dense rounds %.0f | resample %.0f | exit %.0f  (barrier passed %.0f, wave life %.0f)"
          % (c, m.sum(), d[0], d[1], d[2], d[3], d[4] - d[0] - d[1] - d[2] - d[3], d[4], T[m, 1].mean(), T[m, 0].mean()))

E = raw[:, 56:68].copy().view(np.uint32).astype(np.float64)
print("prologue, finer: state decoded %.0f | round trip 2 issued (DMA last) %.0f | pool filed, missiles' segments %.0f | shells done %.0f | background stored %.0f | strokes built+tests %.0f | barrier %.0f"
      % (P[:,0].mean(), P[:,1].mean(), E[:,0].mean(), E[:,1].mean(), E[:,2].mean(), P[:,2].mean(), T[:,1].mean()))

H = raw[:, 68:72].copy().view(np.uint32).astype(np.float64)[:, 0]
f3 = raw[:, 3]
for nm, m in (("text not baked", (f[:,0]&1)==0), ("text baked, bar not", ((f[:,0]&1)==1) & (((f[:,0]>>1)&1)==0)), ("both baked", (f[:,0]&3)==3)):
    print("%-20s n=%5d: shells done %.0f | score block done %.0f | end %.0f   close_text %.2f close_bar %.2f score_pre %.2f bar_pre %.2f" % (
        nm, m.sum(), T[m,3].mean(), H[m].mean(), T[m,0].mean(), (f3[m]&1).mean(), ((f3[m]>>1)&1).mean(), ((f3[m]>>2)&1).mean(), ((f3[m]>>3)&1).mean()))
print("shells merged into the main call: %.3f of the frames with shells" % (((f3>>4)&1)[((f[:,1]>>3)&1)==1].mean()))
