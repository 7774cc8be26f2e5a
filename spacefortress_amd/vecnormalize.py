"""SFVecNormalize -- gym_vecenv.VecNormalize(envs) for the on-device batch (rl/train.py:35-36).

Same constructor arguments and behaviour as the wrapper the trainer applies to 1-D observations
(OpenAI-baselines vintage of gym-vecenv 1.0, see include/sfmi.h); the running statistics live on the
device and are updated and applied by two small HIP kernels right behind sf_step (sf_normalize.hip), so
observations and rewards never visit the host.
"""
import ctypes as C
import collections

import numpy as np
import torch

from . import _lib

RMS = collections.namedtuple("RMS", "mean var count")


class _Params(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("obs_dim", C.c_int32), ("device_id", C.c_int32), ("obs_f64", C.c_int32),
                ("ob", C.c_int32), ("ret", C.c_int32), ("clipob", C.c_double), ("cliprew", C.c_double),
                ("gamma", C.c_double), ("epsilon", C.c_double)]


class SFVecNormalize:
    def __init__(self, venv, ob=True, ret=True, clipob=10., cliprew=10., gamma=0.99, epsilon=1e-8):
        if len(venv.observation_space.shape) != 1:
            raise ValueError("VecNormalize is applied to 1-D observations (rl/train.py:35)")
        self.venv = venv
        self._L = _lib.lib()
        self.num_envs = venv.num_envs
        self.observation_space = venv.observation_space
        self.action_space = venv.action_space
        self.device = venv.device
        self.ob, self.ret_on = bool(ob), bool(ret)
        p = _Params(venv.num_envs, venv.obs_dim, venv.device.index, int(venv.obs_dtype == torch.float64), int(ob), int(ret),
                    clipob, cliprew, gamma, epsilon)
        h = C.c_void_p()
        _lib.check(self._L.sf_normalizer_create(C.byref(p), C.byref(h)))
        self._h = h
        self.training = True
        self._rew = torch.empty(self.num_envs, dtype=torch.float32, device=self.device)
        self._pending = None

    def _stream(self):
        return _lib.raw_stream(self.device)

    def _filter(self, obs, rew):
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        rew_out = self._rew if rew is not None else None
        _lib.check(self._L.sf_normalize(self._h, ptr(obs) if self.ob else None, ptr(obs) if self.ob else None,
                                        ptr(rew) if self.ret_on else None, ptr(rew_out) if self.ret_on else None,
                                        0 if self.training else 1, self._stream()))
        if rew is None:
            return obs, None
        return obs, (rew_out if self.ret_on else rew.float())

    # ------------------------------------------------------------------ VecEnv API
    def reset(self, numpy=False):
        obs = self.venv.reset()
        obs, _ = self._filter(obs, None)
        return obs.cpu().numpy() if numpy else obs

    def step_tensors(self, actions):
        v = self.venv
        if not (self.ob and self.ret_on) or not hasattr(v, "_alloc"):
            obs, rew, done, info = v.step_tensors(actions)
            obs, rew = self._filter(obs, rew)
            return obs, rew, done, info
        # the fused path (sfmi.h: sf_step_normalize): the reduction rides on the step kernel
        if actions.device != v.device or not actions.is_contiguous() or actions.numel() != v.num_envs:
            raise ValueError("actions must be a contiguous tensor of %d elements on %s" % (v.num_envs, v.device))
        at = {torch.uint8: 1, torch.int32: 4, torch.int64: 8}.get(actions.dtype)
        if at is None:
            raise TypeError("actions dtype must be uint8, int32 or int64 (got %s)" % (actions.dtype,))
        obs, rew, done, info = v._alloc()
        p = lambda t: C.c_void_p(t.data_ptr())
        v._before_step(actions)
        _lib.check(self._L.sf_step_normalize(v._h, self._h, p(actions), at, p(obs), p(rew), p(done), p(info), p(self._rew),
                                             0 if self.training else 1, self._stream()))
        v._stepped(actions, rew, done, info)  # (rew: the engine's int reward; self._rew the normalised one)
        return obs, self._rew, done, info

    def step_async(self, actions):
        if torch.is_tensor(actions):
            o, r, d, i = self.step_tensors(actions)
            self._pending = ((o, r, d.view(torch.bool), i.view(torch.bool)), False)
            return
        self.venv.step_async(actions)
        (obs, rew, done, info), _ = self.venv._pending
        self.venv._pending = None
        obs, rew = self._filter(obs, rew)
        self._pending = ((obs, rew, done, info), True)

    def step_wait(self):
        (obs, rew, done, info), as_numpy = self._pending
        self._pending = None
        if not as_numpy:
            return obs, rew, done, info
        return (obs.cpu().numpy(), rew.cpu().numpy().astype(np.float64), done.cpu().numpy().astype(bool),
                info.cpu().numpy().astype(bool))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if getattr(self, "_h", None):
            self._L.sf_normalizer_destroy(self._h)
            self._h = None
        self.venv.close()

    # ------------------------------------------------------------------ statistics
    def _state(self, with_ret=False):
        d = self.venv.obs_dim
        st = np.zeros(2 * d + 4)
        ret = np.zeros(self.num_envs) if with_ret else None
        _lib.check(self._L.sf_normalizer_get_state(self._h, st.ctypes.data_as(C.c_void_p),
                                                   ret.ctypes.data_as(C.c_void_p) if with_ret else None, self._stream()))
        return st, ret

    @property
    def ob_rms(self):
        st, _ = self._state()
        d = self.venv.obs_dim
        return RMS(st[:d].copy(), st[d:2 * d].copy(), st[2 * d + 2])

    @property
    def ret_rms(self):
        st, _ = self._state()
        d = self.venv.obs_dim
        return RMS(st[2 * d], st[2 * d + 1], st[2 * d + 3])

    @property
    def ret(self):
        return self._state(True)[1]

    def state_dict(self):
        st, ret = self._state(True)
        return {"stats": st, "ret": ret}

    def load_state_dict(self, sd):
        st, ret = np.ascontiguousarray(sd["stats"], np.float64), np.ascontiguousarray(sd["ret"], np.float64)
        _lib.check(self._L.sf_normalizer_set_state(self._h, st.ctypes.data_as(C.c_void_p), ret.ctypes.data_as(C.c_void_p),
                                                   self._stream()))
