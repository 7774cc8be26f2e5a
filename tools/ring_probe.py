"""The image step taken apart: render alone / with sf_step, flat output / one slot / rotating slots of the 4-frame ring, with and
without the done flags.   python tools/ring_probe.py   (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacefortress_amd import SFVecEnv, FrameStack
n=16384; steps=500
env = SFVecEnv(n, gametype="youturn", obs_type="image", spawn_stride=1, reuse_buffers=True)
st = FrameStack(env, 4); st.reset()
acts = torch.randint(0, env.n_actions, (64, n), device=env.device, dtype=torch.uint8)
for t in range(400): st.step(acts[t % 64])
torch.cuda.synchronize()
def timeit(f):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(20): f(i)
    e0.record()
    for i in range(steps): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/steps*1e3
flat = torch.empty((n,1,84,84), dtype=torch.uint8, device=env.device)
print("render alone, flat [N,1,84,84]          %.1f us" % timeit(lambda i: env.render("image", out=flat)))
print("render alone, one slot of the ring      %.1f us" % timeit(lambda i: env.render("image", out=st._slot(1))))
print("render alone, rotating slots of the ring %.1f us" % timeit(lambda i: env.render("image", out=st._slot(i % 4))))
import ctypes as C
from spacefortress_amd import _lib
done = torch.zeros(n, dtype=torch.uint8, device=env.device)
def rs(i):
    _lib.check(env._L.sf_render_stack(env._h, C.c_void_p(st.ring.data_ptr()), 4, i % 4, C.c_void_p(done.data_ptr()), env._stream()))
print("sf_render_stack alone, rotating, done=0  %.1f us" % timeit(rs))
def rs1(i):
    _lib.check(env._L.sf_render_stack(env._h, C.c_void_p(st.ring.data_ptr()), 4, 1, C.c_void_p(done.data_ptr()), env._stream()))
print("sf_render_stack alone, one slot, done=0  %.1f us" % timeit(rs1))
rew = torch.empty(n, dtype=torch.int32, device=env.device); dn = torch.empty(n, dtype=torch.uint8, device=env.device); inf = torch.empty(n, dtype=torch.uint8, device=env.device)
def stp(i):
    a = acts[i % 64]
    _lib.check(env._L.sf_step(env._h, C.c_void_p(a.data_ptr()), 1, None, C.c_void_p(rew.data_ptr()), C.c_void_p(dn.data_ptr()), C.c_void_p(inf.data_ptr()), env._stream()))
def a1(i): stp(i); env.render("image", out=flat)
def a2(i): stp(i); env.render("image", out=st._slot(i % 4))
def a3(i): stp(i); _lib.check(env._L.sf_render_stack(env._h, C.c_void_p(st.ring.data_ptr()), 4, i % 4, C.c_void_p(done.data_ptr()), env._stream()))
def a4(i): stp(i); _lib.check(env._L.sf_render_stack(env._h, C.c_void_p(st.ring.data_ptr()), 4, i % 4, C.c_void_p(dn.data_ptr()), env._stream()))
def a5(i): stp(i); _lib.check(env._L.sf_render_stack(env._h, C.c_void_p(st.ring.data_ptr()), 4, 1, C.c_void_p(dn.data_ptr()), env._stream()))
print("step only                                          %.1f us" % timeit(stp))
print("step + render flat                                 %.1f us" % timeit(a1))
print("step + render rotating slots (no done)             %.1f us" % timeit(a2))
print("step + sf_render_stack rotating, done = zeros      %.1f us" % timeit(a3))
print("step + sf_render_stack rotating, done of the step  %.1f us" % timeit(a4))
print("step + sf_render_stack one slot, done of the step  %.1f us" % timeit(a5))
ring2 = torch.empty((4, n, 1, 84, 84), dtype=torch.uint8, device=env.device)
def b1(i): stp(i); env.render("image", out=ring2[i % 4])
print("step + render rotating, SLOT-MAJOR ring [4][N]        %.1f us" % timeit(b1))
print("render alone rotating, SLOT-MAJOR ring [4][N]         %.1f us" % timeit(lambda i: env.render("image", out=ring2[i % 4])))
big = torch.empty((16, n, 1, 84, 84), dtype=torch.uint8, device=env.device)
def b2(i): stp(i); env.render("image", out=big[i % 16])
print("step + render rotating over 16 flat buffers (1.8 GB)  %.1f us" % timeit(b2))
two = torch.empty((2, n, 1, 84, 84), dtype=torch.uint8, device=env.device)
def b3(i): stp(i); env.render("image", out=two[i % 2])
print("step + render rotating over 2 flat buffers (231 MB)   %.1f us" % timeit(b3))
