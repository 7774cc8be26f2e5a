import ctypes as C, os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/spacefortress_amd") else os.getcwd())
os.environ["SFMI_LIB_PATH"] = os.path.abspath("build/diag/libsfmi_rep.so")
import numpy as np, torch
from spacefortress_amd import SFVecEnv, _lib
n = 65536
env = SFVecEnv(n, gametype="youturn", spawn_stride=1, reuse_buffers=True)
acts = torch.randint(0, env.n_actions, (64, n), device=env.device, dtype=torch.uint8)
for t in range(300):
    env.step_tensors(acts[t % 64])
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros((n // 64, 16), np.uint64)
L.sf_debug_read.argtypes = [C.c_void_p, C.c_void_p]
L.sf_debug_read(env._h, buf.ctypes.data_as(C.c_void_p))
s = buf.astype(np.int64)
print("rep0 start -> rep1 start (cold): median %.0f cycles" % np.median(s[:, 11] - s[:, 10]))
print("rep1 start -> end of rep2 (2 warm reps): median %.0f cycles, per rep %.0f" % (np.median(s[:, 8] - s[:, 11]), np.median(s[:, 8] - s[:, 11]) / 2))
print("kernel entry -> rep0 start: %.0f" % np.median(s[:, 10] - s[:, 0]))
