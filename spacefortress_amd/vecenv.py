"""SFVecEnv -- the on-device batch that stands in for gym_vecenv.SubprocVecEnv.

The reference runs N `SSF_Env`s in N processes and the trainer talks to them as
one VecEnv (rl/train.py:30-41,60,80-85,179):

    envs.observation_space.shape, envs.action_space
    obs = envs.reset()                        -> [N, ...]
    obs, reward, done, info = envs.step(a)    -> [N, ...], [N], [N], N bools
    envs.close()

SFVecEnv keeps that surface.  The N environments are lanes of one HIP kernel
(libsfmi.so, include/sfmi.h); PyTorch only owns the I/O buffers and the stream.
`step` accepts a CUDA/HIP tensor (returns tensors, nothing leaves the device,
nothing synchronises) or a numpy array / list (returns numpy arrays, like the
reference).  Auto-reset on `done` is the worker loop's: the observation of a
finished lane is the first one of its next episode, reward/done/info belong to
the finished step.

There is no CPU fallback: without a GPU (or without libsfmi.so) construction raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .spaces import Box, Discrete

_ACT_TYPES = {torch.uint8: 1, torch.int32: 4, torch.int64: 8}
_NP_FIELD_DTYPES = {(1, 0): np.uint8, (2, 0): np.int16, (4, 0): np.int32, (8, 0): np.int64,
                    (4, 1): np.float32, (8, 1): np.float64}
_UNSIGNED_FIELDS = {"spawn_cursor", "missile_mask", "shell_mask", "flags"}


class SFVecEnv:
    def __init__(self, num_envs, gametype="youturn", obs_type="features", action_set=1, device=None,
                 seed=1, spawn_skip=0, spawn_stride=0, obs_dtype=torch.float32, faithful_bugs=True,
                 auto_reset=True, spawn_table_len=0, reuse_buffers=False, image_geometry=None, ref_reset_obs=False):
        if obs_type not in _lib.OBS_TYPES:
            raise AssertionError("obs_type %r" % (obs_type,))  # ENV:51
        self._L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.SfmiError("SFVecEnv needs a HIP device (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise _lib.SfmiError("SFVecEnv runs on the GPU only (got device %s)" % (self.device,))
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        if obs_dtype not in (torch.float32, torch.float64):
            raise ValueError("obs_dtype must be torch.float32 or torch.float64")
        flags = 0
        if obs_dtype == torch.float64:
            flags |= _lib.FLAG_OBS_F64
        if not faithful_bugs:
            flags |= _lib.FLAG_REAL_SHELL_COUNT
        if not auto_reset:
            flags |= _lib.FLAG_NO_AUTO_RESET
        if ref_reset_obs:  # a new game's observation as the reference's wrapper returns it: aim = vdir = ndist = 0 (sfmi.h)
            flags |= _lib.FLAG_REF_RESET_OBS
        p = _lib.CreateParams(gametype.encode(), int(num_envs), dev_index, int(action_set),
                              _lib.OBS_TYPES[obs_type], flags, int(seed) & 0xFFFFFFFF, int(spawn_skip),
                              int(spawn_stride), int(spawn_table_len))
        h = C.c_void_p()
        _lib.check(self._L.sf_create(C.byref(p), C.byref(h)))
        self._h = h
        # what a replay file needs to make this batch again (spacefortress_amd/replay.py)
        self._create = {"action_set": int(action_set), "seed": int(seed) & 0xFFFFFFFF, "spawn_skip": int(spawn_skip),
                        "spawn_stride": int(spawn_stride), "auto_reset": bool(auto_reset)}
        self.default_geometry = True
        self._durations = None
        self._fresh = True  # nothing has changed the state sf_create left: a recording may start here
        self._rec = None
        self.num_envs = int(num_envs)
        self.gametype = gametype
        self.obs_type = obs_type
        self.obs_dtype = obs_dtype
        self.obs_dim = self._L.sf_obs_dim(h)
        # 'image': what the trainer's VecEnv yields, WrapPyTorch's [1, 84, 84] uint8 (rl/envs.py:19-30);
        # 'image-raw': SSF_Env's own [92, 90] grey frame (ENV:171)
        self.is_image = obs_type in ("image", "image-raw")
        self.image_w, self.image_h = _lib.IMAGE_W, _lib.IMAGE_H
        self.obs_shape = {"image": (1, _lib.IMAGE_OUT, _lib.IMAGE_OUT),
                          "image-raw": (self.image_h, self.image_w)}.get(obs_type, (self.obs_dim,))
        self.n_actions = self._L.sf_n_actions(h)
        self.tickdur = self._L.sf_tick_ms(h)      # ENV:61
        self.max_ticks = self._L.sf_max_ticks(h)  # ENV:165
        self.action_space = Discrete(self.n_actions)  # ENV:90
        # ENV:175 declares dtype uint8 for the feature Box; the values are floats
        if self.is_image:
            self.obs_dtype = torch.uint8
            self.observation_space = Box(0, 255, self.obs_shape, np.uint8)  # rl/envs.py:22-26
        else:
            self.observation_space = Box(-np.inf, np.inf, (self.obs_dim,),
                                         np.float32 if obs_dtype == torch.float32 else np.float64)
        self.reuse_buffers = reuse_buffers
        self._bufs = None
        self._buf_ptrs = None
        self._pending = None
        self._fields = None
        # SSF_Env(scale, viewport, ls) (ENV:50-60): the frames' geometry; None = the reference's default (.2, (130, 80, 450, 460), 3)
        if image_geometry is not None:
            try:
                self.set_image_geometry(*image_geometry)
            except Exception:
                self.close()
                raise

    # ------------------------------------------------------------------ buffers
    def _stream(self):
        return _lib.raw_stream(self.device)

    def _alloc(self):
        if self.reuse_buffers and self._bufs is not None:
            return self._bufs
        n = self.num_envs
        # reward, done and info share one allocation (4 n + n + n bytes): the host API brings them over in ONE copy
        rdi = torch.empty(6 * n, dtype=torch.uint8, device=self.device)
        bufs = (torch.empty((n,) + self.obs_shape, dtype=self.obs_dtype, device=self.device),
                rdi[:4 * n].view(torch.int32), rdi[4 * n:5 * n], rdi[5 * n:])
        if self.reuse_buffers:
            self._bufs = bufs
            self._buf_ptrs = tuple(C.c_void_p(t.data_ptr()) for t in bufs)
        return bufs

    # ------------------------------------------------------------------ VecEnv API
    def reset(self, numpy=False):
        """env.reset() in every lane (ENV:163-178): new games; returns obs [N, obs_dim]."""
        obs = self._alloc()[0]
        self._touch()  # (new Games: a recording ends, the duration log's vectors start empty)
        _lib.check(self._L.sf_reset(self._h, C.c_void_p(obs.data_ptr()), self._stream()))
        return obs.cpu().numpy() if numpy else obs

    def step_tensors(self, actions, out=None):
        """Device fast path: `actions` is a contiguous uint8/int32/int64 tensor on this device.
        Returns (obs, reward int32, done uint8, info uint8) tensors; nothing synchronises."""
        if actions.device != self.device or not actions.is_contiguous() or actions.numel() != self.num_envs:
            raise ValueError("actions must be a contiguous tensor of %d elements on %s" % (self.num_envs, self.device))
        at = _ACT_TYPES.get(actions.dtype)
        if at is None:
            raise TypeError("actions dtype must be uint8, int32 or int64 (got %s)" % (actions.dtype,))
        if out is None and self._bufs is not None and self.reuse_buffers:  # the addresses of the reused outputs, made once
            bufs, ptrs = self._bufs, self._buf_ptrs
        else:
            bufs = out if out is not None else self._alloc()
            ptrs = tuple(C.c_void_p(t.data_ptr()) for t in bufs)
        self._before_step(actions)
        _lib.check(self._L.sf_step(self._h, C.c_void_p(actions.data_ptr()), at, ptrs[0], ptrs[1], ptrs[2], ptrs[3],
                                   self._stream()))
        self._stepped(actions, bufs[1], bufs[2], bufs[3])
        return bufs

    def rollout(self, actions, out=None, want_obs=True):
        """K steps whose actions are all known up front, fused into one launch (sfmi.h: sf_rollout).
        `actions`: contiguous uint8/int32/int64 tensor [K, N] on this device.  Returns
        (obs [K, N, obs_dim] or None, reward int32 [K, N], done uint8 [K, N], info uint8 [K, N]);
        bit-identical to K `step_tensors` calls.  (An image batch with want_obs gets its frames [K, N, h, w] from K step
        launches each followed by its frames -- the fused launch keeps the state in registers --, in the one call.)"""
        if actions.device != self.device or not actions.is_contiguous() or actions.dim() != 2 \
                or actions.shape[1] != self.num_envs:
            raise ValueError("actions must be a contiguous [K, %d] tensor on %s" % (self.num_envs, self.device))
        at = _ACT_TYPES.get(actions.dtype)
        if at is None:
            raise TypeError("actions dtype must be uint8, int32 or int64 (got %s)" % (actions.dtype,))
        K, n = actions.shape
        if out is not None:
            obs, rew, done, info = out
        else:
            obs = torch.empty((K, n) + self.obs_shape, dtype=self.obs_dtype, device=self.device) if want_obs else None
            rew = torch.empty((K, n), dtype=torch.int32, device=self.device)
            done = torch.empty((K, n), dtype=torch.uint8, device=self.device)
            info = torch.empty((K, n), dtype=torch.uint8, device=self.device)
        self._no_duration_log("rollout()")
        ev = self._rollout_events_begin(K)
        try:
            _lib.check(self._L.sf_rollout(self._h, C.c_void_p(actions.data_ptr()), at, int(K),
                                          C.c_void_p(obs.data_ptr()) if obs is not None else None,
                                          C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
                                          C.c_void_p(info.data_ptr()), self._stream()))
        finally:
            self._rollout_events_end(ev)
        self._fresh = False
        if self._rec is not None:
            self._rec.add(actions.to(torch.uint8), rew, done, info)
        return obs, rew, done, info

    # ------------------------------------------------------------------ replay files (spacefortress_amd/replay.py)
    def _before_step(self, actions):
        """THE hook in FRONT of every single-step launch with given actions -- step_tensors and the wrappers that call the C
        ABI themselves (FrameStack.step, DeviceRollout.step, SFVecNormalize's fused step): the duration log reads the flags,
        timers and vulnerability the tick's key calls meet (durations.py) before the launch changes them."""
        if self._durations is not None:
            self._durations.before(actions)

    def _no_duration_log(self, what):
        """Launches that play several ticks, or actions nobody has seen yet (rollout, the *_sampled calls), give the duration
        log nothing to read between the ticks: refused while one is enabled, rather than leaving its vectors stale."""
        if self._durations is not None:
            raise RuntimeError("%s while a duration log is enabled (enable_durations): the log follows single steps with "
                               "given actions -- step / step_tensors and the wrappers built on them" % what)

    def _stepped(self, actions, rew, done, info):
        """THE hook behind every launch that steps the batch -- step_tensors / rollout / the *_sampled calls above, and the
        wrappers that call the C ABI themselves (FrameStack.step, DeviceRollout.step, SFVecNormalize's fused step): the state is
        no longer the one sf_create left, the duration log appends what the tick pushed (and empties the vectors of an env
        whose episode ended), and a recording gets the actions with the engine's own (unnormalised) reward, done and info of
        the step(s): [N] or [K, N] device tensors, not synchronised."""
        self._fresh = False
        if self._durations is not None:
            self._durations.after(done)
        if self._rec is not None:
            self._rec.add(actions.to(torch.uint8), rew, done, info)

    def _touch(self):
        """The state is about to change otherwise than by a recorded step: a recording cannot go on, and the duration log starts
        over (its vectors belong to the Games the batch held: reset() makes new ones, set_field / load_state_dict put the batch
        somewhere the log has not followed)."""
        self._fresh = False
        if self._durations is not None:
            self._durations.reset()
        if self._rec is not None:
            self._rec = None
            raise RuntimeError("reset() / set_field() during a recording: a replay file is the game from a NEW batch on; "
                               "the recording was dropped")

    def start_recording(self):
        """From here on every action played is kept (device tensors, no synchronisation); save_replay() writes them out with
        what they produced.  Only on a batch nothing has stepped or reset since it was made: a replay starts from the state
        sf_create leaves (the reference has no way to put a Game into a given state either)."""
        from .replay import Recorder

        if not self._fresh:
            raise RuntimeError("start_recording() needs a new batch (nothing stepped, reset or edited since SFVecEnv(...))")
        self._rec = Recorder(self)

    def save_replay(self, path):
        """Write the recording so far (gametype, action set, seed, spawn offsets, build id, uint8 [T, N] actions, per-env
        returns / kills / episode ends, digest of the state now) to `path`; the recording goes on."""
        if self._rec is None:
            raise RuntimeError("save_replay(): no recording (start_recording() on a new batch first)")
        rp = self._rec.replay(self)
        rp.save(path)
        return rp

    @staticmethod
    def load_replay(path):
        """The file as a spacefortress_amd.replay.Replay: `.run()` plays it on the device and verifies it."""
        from .replay import Replay

        return Replay.load(path)

    def _rollout_events_begin(self, K):
        """A fused launch stores one row of event masks PER TICK (a.events[step * n_envs + i], sf_kernels.hip): with
        events enabled it gets a [K, N] buffer of its own for the duration of the launch (`self.rollout_events`);
        `self.events` holds one tick's row and would be overrun by K - 1 rows."""
        ev = None
        if getattr(self, "events", None) is not None:
            ev = torch.zeros((K, self.num_envs), dtype=torch.int32, device=self.device)
            _lib.check(self._L.sf_set_event_output(self._h, C.c_void_p(ev.data_ptr())))
        self.rollout_events = ev
        return ev

    def _rollout_events_end(self, ev):
        if ev is not None:
            _lib.check(self._L.sf_set_event_output(self._h, C.c_void_p(self.events.data_ptr())))

    def seed_actions(self, seed, first_lane=0):
        """Restart the on-device action sampler of `step_sampled` at tick 0 (sfmi.h: sf_seed_actions): lane i then plays
        Philox4x32-10(key = seed, counter = (first_lane + i, tick)) scaled to [0, n_actions).  `first_lane` = this shard's
        first lane in a job of several batches, so that shards draw what the one big batch would."""
        _lib.check(self._L.sf_seed_actions(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_lane) & 0xFFFFFFFF, self._stream()))

    def step_sampled(self, out=None, actions_out=None):
        """One step on actions the lanes draw themselves inside the launch (sfmi.h: sf_step_sampled) -- the random-action
        rollout without an action tensor.  Returns (obs, reward, done, info) like `step_tensors`; `actions_out` (uint8 [N]
        on this device, optional) receives what was played."""
        if out is None and self._bufs is not None and self.reuse_buffers:
            bufs, ptrs = self._bufs, self._buf_ptrs
        else:
            bufs = out if out is not None else self._alloc()
            ptrs = tuple(C.c_void_p(t.data_ptr()) for t in bufs)
        ao = None
        if actions_out is None and self._rec is not None:  # a recording wants to know what was drawn
            actions_out = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        if actions_out is not None:
            if actions_out.device != self.device or actions_out.dtype != torch.uint8 or not actions_out.is_contiguous() \
                    or actions_out.numel() != self.num_envs:
                raise ValueError("actions_out must be a contiguous uint8 tensor of %d elements on %s" % (self.num_envs, self.device))
            ao = C.c_void_p(actions_out.data_ptr())
        self._no_duration_log("step_sampled()")
        _lib.check(self._L.sf_step_sampled(self._h, ao, ptrs[0], ptrs[1], ptrs[2], ptrs[3], self._stream()))
        self._fresh = False
        if self._rec is not None:
            self._rec.add(actions_out, bufs[1], bufs[2], bufs[3])
        return bufs

    def rollout_sampled(self, n_steps, want_obs=True, want_actions=True):
        """`rollout` on sampled actions: K ticks in one launch.  Returns (obs, reward, done, info, actions uint8 [K, N])."""
        K, n = int(n_steps), self.num_envs
        self._no_duration_log("rollout_sampled()")
        obs = torch.empty((K, n) + self.obs_shape, dtype=self.obs_dtype, device=self.device) if want_obs else None
        rew = torch.empty((K, n), dtype=torch.int32, device=self.device)
        done = torch.empty((K, n), dtype=torch.uint8, device=self.device)
        info = torch.empty((K, n), dtype=torch.uint8, device=self.device)
        acts = torch.empty((K, n), dtype=torch.uint8, device=self.device) if (want_actions or self._rec is not None) else None
        ev = self._rollout_events_begin(K)
        try:
            _lib.check(self._L.sf_rollout_sampled(self._h, K, C.c_void_p(acts.data_ptr()) if acts is not None else None,
                                                  C.c_void_p(obs.data_ptr()) if obs is not None else None,
                                                  C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()),
                                                  C.c_void_p(info.data_ptr()), self._stream()))
        finally:
            self._rollout_events_end(ev)
        self._fresh = False
        if self._rec is not None:
            self._rec.add(acts, rew, done, info)
        return obs, rew, done, info, acts

    def step_async(self, actions):
        if torch.is_tensor(actions):
            self._pending = (self.step_tensors(actions), False)
            return
        a = np.asarray(actions)
        if a.shape != (self.num_envs,):
            a = a.reshape(self.num_envs)
        if a.size and (a.min() < 0 or a.max() >= self.n_actions):
            # ENV:211-212: actions_taken[action] / action_combinations[action]
            raise IndexError("action out of range for Discrete(%d)" % self.n_actions)
        t = torch.from_numpy(np.ascontiguousarray(a, np.int64)).to(self.device)
        self._pending = (self.step_tensors(t), True)

    def step_wait(self):
        (obs, rew, done, info), as_numpy = self._pending
        self._pending = None
        if not as_numpy:
            return obs, rew, done.view(torch.bool), info.view(torch.bool)  # 0 / 1 bytes: views, not kernels
        # np.stack of per-env python ints / bools, as the subprocess vec-env returns them
        base, n = done._base, self.num_envs
        if base is not None and info._base is base and base.numel() == 6 * n and rew.data_ptr() == base.data_ptr():
            h = base.cpu().numpy()  # one transfer for the three small results (_alloc)
            return (obs.cpu().numpy(), h[:4 * n].view(np.int32).astype(np.int64), h[4 * n:5 * n].astype(bool),
                    h[5 * n:].astype(bool))
        return (obs.cpu().numpy(), rew.cpu().numpy().astype(np.int64), done.cpu().numpy().astype(bool),
                info.cpu().numpy().astype(bool))

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._L.sf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_image_geometry(self, scale=.2, viewport=(130, 80, 450, 460), ls=3):
        """The geometry of this batch's frames, as SSF_Env's constructor takes it (ENV:50-60; sfmi.h: sf_set_image_geometry).
        The default has the fast frame kernel; any other one the general renderer (surface width and height in [84, 251])."""
        vx, vy, vw, vh = (float(v) for v in viewport)
        _lib.check(self._L.sf_set_image_geometry(self._h, float(scale), vx, vy, vw, vh, float(ls)))
        w, h = C.c_int32(), C.c_int32()
        _lib.check(self._L.sf_image_size(self._h, C.byref(w), C.byref(h)))
        self.image_w, self.image_h = int(w.value), int(h.value)
        # (the default geometry's frame kernel also shifts a frame stack in the same launch -- sf_render_shift; the general
        #  renderer draws into one slot: DeviceRollout asks)
        self.default_geometry = bool(self._L.sf_image_geometry_is_default(self._h))  # the library's word, not a second predicate
        if self.obs_type == "image-raw":
            self.obs_shape = (self.image_h, self.image_w)
            self.obs_dim = self._L.sf_obs_dim(self._h)
            self.observation_space = Box(0, 255, self.obs_shape, np.uint8)
            self._bufs = self._buf_ptrs = None

    def set_score_glyphs(self, alpha=None, layout=None, x0=None):
        """The glyph atlas the score text is drawn from in the batch's CURRENT geometry (sfmi.h: sf_set_score_glyphs; how one is
        taken from a box's cairo: tests/golden/frames/make_score_golden.py).  alpha uint8 [11, gh, gw] ('0'..'9', '-'), layout
        (gw, gh, advance, y0), x0 int [11, 10] (or one int).  alpha=None: the seven-segment fallback, by name.  The default
        geometry starts with the built-in atlas (= the reference's frames on this image), any other geometry with none."""
        if alpha is None:
            _lib.check(self._L.sf_set_score_glyphs(self._h, None, None))
            return
        a = np.ascontiguousarray(alpha, np.uint8)
        gw, gh, adv, y0 = (int(v) for v in layout)
        if a.shape != (11, gh, gw):
            raise ValueError("alpha must be uint8 [11, gh, gw] = [11, %d, %d]" % (gh, gw))
        g = _lib.ScoreGlyphs(gw, gh, adv, y0)
        xs = np.broadcast_to(np.asarray(x0, np.int16), (11, 10))
        for i in range(11):
            for j in range(10):
                g.x0[i][j] = int(xs[i, j])
        _lib.check(self._L.sf_set_score_glyphs(self._h, C.byref(g), a.ctypes.data_as(C.c_void_p)))

    def score_glyphs(self):
        """The atlas in use: dict(alpha, layout, x0), or None while the seven-segment fallback draws the text."""
        has, g = C.c_int32(), _lib.ScoreGlyphs()
        a = np.zeros(11 * 24 * 24, np.uint8)
        _lib.check(self._L.sf_get_score_glyphs(self._h, C.byref(has), C.byref(g), a.ctypes.data_as(C.c_void_p), a.size))
        if not has.value:
            return None
        return dict(alpha=a[:11 * g.gw * g.gh].reshape(11, g.gh, g.gw).copy(), layout=(g.gw, g.gh, g.advance, g.y0),
                    x0=np.array([[g.x0[i][j] for j in range(10)] for i in range(11)], np.int16))

    def render(self, mode="image-raw", out=None):
        """Frames of the CURRENT state of every env, whatever obs_type the batch steps with (the
        reference's `render()`, ENV:180-198): uint8 [N, 92, 90] ('image-raw') or [N, 1, 84, 84] ('image')."""
        if mode not in ("image", "image-raw"):
            raise ValueError("mode must be 'image' or 'image-raw'")
        shape = (1, _lib.IMAGE_OUT, _lib.IMAGE_OUT) if mode == "image" else (self.image_h, self.image_w)
        if out is None:
            out = torch.empty((self.num_envs,) + shape, dtype=torch.uint8, device=self.device)
        # `out` may be a view whose env stride is larger than a frame (one slot of a frame stack)
        if out.shape != (self.num_envs,) + shape or out.dtype != torch.uint8 or not out[0].is_contiguous():
            raise ValueError("out must be uint8 %s with contiguous frames" % (((self.num_envs,) + shape),))
        stride = out.stride(0) if self.num_envs > 1 else 0
        _lib.check(self._L.sf_render(self._h, _lib.OBS_TYPES[mode], C.c_void_p(out.data_ptr()), stride, self._stream()))
        return out

    def draw_records(self, from_state=False):
        """Diagnostics: the envs' draw records (sfmi.h: sf_draw_records) as uint8 [N, 432] -- what the frame kernel reads
        instead of the state.  from_state=True rebuilds them from the state first (what a frame does after reset() /
        set_field()); False returns what the last step launch of an image batch left."""
        out = np.empty((self.num_envs, 432), np.uint8)
        _lib.check(self._L.sf_draw_records(self._h, out.ctypes.data_as(C.c_void_p), out.nbytes, int(bool(from_state))))
        return out

    def enable_durations(self, capacity=512):
        """The reference's four telemetry vectors (SRC/game.hh:98-101; getters SRC/pymodule.cpp:143-181) for every env of the
        batch: `thrust_durations`, `shot_durations`, `shot_intervals_invul`, `shot_intervals_vul`, kept on the device by
        `step_tensors` / `step` from here on (durations.py).  Returns the log: `log.get(name)` -> (values [N, capacity],
        counts [N]).  Off by default: four small gathers and a handful of elementwise kernels per step."""
        from .durations import DurationLog
        self._durations = DurationLog(self, capacity)
        return self._durations

    def enable_events(self, on=True):
        """Per-tick event bitmasks (sfmi.h SF_EV_*; `_lib.EVENT_NAMES`): after every step `self.events` holds
        uint32 [N] for that tick.  Off by default: it is one more 4-byte store per env and step."""
        self.events = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device) if on else None
        _lib.check(self._L.sf_set_event_output(self._h, C.c_void_p(self.events.data_ptr()) if on else None))
        return self.events

    # ------------------------------------------------------------------ extras
    def check_actions(self):
        """Raise IndexError if any action since the last call was out of range (device path)."""
        _lib.check(self._L.sf_check_actions(self._h, self._stream()))

    def check_state(self):
        """Raise OverflowError if a per-episode counter or key timer outgrew its packed width since the last call
        (sfmi.h: sf_check_state): only a batch with auto_reset=False that is stepped for several episodes' worth of
        ticks without reset() can get there; the reference's plain ints keep counting (SRC/game.hh:29-43)."""
        _lib.check(self._L.sf_check_state(self._h, self._stream()))

    def episode_stats(self, clear=False):
        """Device-accumulated episode statistics as an int64 tensor of 8 (see sfmi.h)."""
        out = np.zeros(_lib.EPISODE_STATS_LEN, np.int64)
        _lib.check(self._L.sf_episode_stats(self._h, out.ctypes.data_as(C.c_void_p), int(clear), self._stream()))
        return out

    def field_names(self):
        if self._fields is None:
            self._fields = {}
            d = _lib.FieldDesc()
            for f in range(self._L.sf_n_fields()):
                self._L.sf_field_info(f, C.byref(d))
                name = d.name.decode()
                dt = _NP_FIELD_DTYPES[(d.elem_size, d.is_float)]
                if name in _UNSIGNED_FIELDS:
                    dt = {1: np.uint8, 4: np.uint32}[d.elem_size]
                elif d.elem_size == 1:
                    dt = np.int8
                self._fields[name] = (f, dt, d.count)
        return list(self._fields)

    def get_field(self, name):
        """One state field as numpy: [N] or [count, N] (slot-major)."""
        self.field_names()
        if name not in self._fields:
            raise KeyError(name)
        f, dt, count = self._fields[name]
        arr = np.empty((count, self.num_envs), dt)
        _lib.check(self._L.sf_get_field(self._h, f, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return arr[0] if count == 1 else arr

    def get_field_tensor(self, name, out=None):
        """One state field as a DEVICE tensor, [N] or [count, N], without synchronising (sfmi.h: sf_get_field_dev): ordered on
        the current stream behind the steps issued there.  Every field but missile_x / missile_y / missile_angle."""
        self.field_names()
        if name not in self._fields:
            raise KeyError(name)
        f, dt, count = self._fields[name]
        tdt = {np.dtype(np.int32): torch.int32, np.dtype(np.uint32): torch.int32, np.dtype(np.float64): torch.float64,
               np.dtype(np.float32): torch.float32, np.dtype(np.int16): torch.int16, np.dtype(np.uint8): torch.uint8,
               np.dtype(np.int8): torch.int8, np.dtype(np.uint64): torch.int64, np.dtype(np.int64): torch.int64}[np.dtype(dt)]
        if out is None:
            out = torch.empty((count, self.num_envs), dtype=tdt, device=self.device)
        _lib.check(self._L.sf_get_field_dev(self._h, f, C.c_void_p(out.data_ptr()), out.numel() * out.element_size(), self._stream()))
        return out[0] if count == 1 and out.dim() == 2 else out

    def set_field(self, name, value):
        self.field_names()
        if name not in self._fields:
            raise KeyError(name)
        f, dt, count = self._fields[name]
        arr = np.ascontiguousarray(np.asarray(value, dt).reshape(count, self.num_envs))
        self._touch()
        _lib.check(self._L.sf_set_field(self._h, f, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def state_dict(self):
        """Every state field (checkpoint / golden replay)."""
        return {n: self.get_field(n) for n in self.field_names()}

    def load_state_dict(self, sd):
        for n in self.field_names():
            self.set_field(n, sd[n])
        # every packed field of every env has just been rewritten with values that fit (set_field range-checks): whatever had
        # wrapped is repaired, the sticky count of check_state() starts over (sfmi.h: sf_clear_state_errors)
        _lib.check(self._L.sf_clear_state_errors(self._h))
