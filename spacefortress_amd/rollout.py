"""DeviceRollout -- RolloutStorage (rl/storage.py:9-64) plus the trainer's per-step bookkeeping
(rl/train.py:74-98), resident on the device.

The reference moves every step through the host: `action.cpu().numpy()` -> SubprocVecEnv pipes ->
`torch.from_numpy(reward)`, five small tensor ops for the episode bookkeeping, `rollouts.insert(...)`.
Here ONE launch per step (`sf_step_record`) writes the next observation straight into
`observations[step + 1]` and, in the step kernel's own epilogue, `rewards[step]`, `masks[step + 1]`, the
episode / final reward accumulators and `actions[step]`; `compute_returns` is one backward-scan kernel
(`sf_compute_returns`, bit-identical to the reference's float32 arithmetic).  Tensor names and shapes are
RolloutStorage's.
"""
import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DeviceRollout:
    def __init__(self, env, num_steps, state_size=1, num_stack=1):
        # `env`: an SFVecEnv, or an SFVecNormalize around one (the trainer's configuration for 1-D observations,
        # rl/train.py:35-36: observations and rewards reach the storage normalised)
        self.norm = env if hasattr(env, "venv") else None
        env = env.venv if self.norm is not None else env
        self.env = env
        self._L = _lib.lib()
        T, n, dev = int(num_steps), env.num_envs, env.device
        self.num_steps = T
        obs_shape = tuple(env.obs_shape)
        # image observations: obs_shape = (obs_shape[0] * num_stack, ...) as in rl/train.py:38-39; every step stores
        # the whole stack (shifted by one frame, zeroed for finished envs, new frame last)
        self.num_stack = int(num_stack)
        if self.num_stack > 1:
            if env.obs_type != "image" or self.norm is not None:
                raise ValueError("num_stack > 1 is for obs_type='image' (rl/train.py:38-39)")
            obs_shape = (self.num_stack,) + obs_shape[1:]
        self.observations = torch.zeros((T + 1, n) + obs_shape, dtype=env.obs_dtype, device=dev)
        self.states = torch.zeros(T + 1, n, state_size, device=dev)
        self.rewards = torch.zeros(T, n, 1, device=dev)
        self.value_preds = torch.zeros(T + 1, n, 1, device=dev)
        self.returns = torch.zeros(T + 1, n, 1, device=dev)
        self.action_log_probs = torch.zeros(T, n, 1, device=dev)
        self.actions = torch.zeros(T, n, 1, dtype=torch.long, device=dev)  # Discrete (rl/storage.py:17-23)
        self.masks = torch.ones(T + 1, n, 1, device=dev)
        self.episode_rewards = torch.zeros(n, 1, device=dev)  # rl/train.py:64-65
        self.final_rewards = torch.zeros(n, 1, device=dev)
        self._kills0 = int(env.episode_stats()[3])
        self._rew = torch.empty(n, dtype=torch.int32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._info = torch.empty(n, dtype=torch.uint8, device=dev)
        self._ptr = None

    def _stream(self):
        return _lib.raw_stream(self.env.device)

    def reset(self):
        """obs = envs.reset(); rollouts.observations[0].copy_(obs) (rl/train.py:60-62)."""
        e = self.env
        e._touch()  # (a recording cannot go on across a reset: vecenv.py)
        if self.num_stack > 1:  # update_current_obs on a zeroed stack (rl/train.py:43,60-62)
            _lib.check(e._L.sf_reset(e._h, None, e._stream()))
            self.observations[0].zero_()
            _lib.check(e._L.sf_render_stack(e._h, _p(self.observations[0]), self.num_stack, self.num_stack - 1, None,
                                            e._stream()))
            return self.observations[0]
        _lib.check(e._L.sf_reset(e._h, _p(self.observations[0]), e._stream()))
        if self.norm is not None:
            self.norm._filter(self.observations[0], None)  # VecNormalize.reset: update + normalise in place
        return self.observations[0]

    def _pointers(self):
        """Raw addresses of every per-step slice, made once: the step loop is host-bound otherwise
        (a tensor view and its data_ptr() cost about a microsecond each)."""
        T = self.num_steps
        vp = C.c_void_p
        self._ptr = {
            "obs": [vp(self.observations[t].data_ptr()) for t in range(T + 1)],
            "rew": [vp(self.rewards[t].data_ptr()) for t in range(T)],
            "mask": [vp(self.masks[t].data_ptr()) for t in range(T + 1)],
            "act": [vp(self.actions[t].data_ptr()) for t in range(T)],
            "r": vp(self._rew.data_ptr()), "d": vp(self._done.data_ptr()), "i": vp(self._info.data_ptr()),
            "ep": vp(self.episode_rewards.data_ptr()), "fin": vp(self.final_rewards.data_ptr()),
        }
        # what step() returns, as views made once
        self._views = [(self.observations[t + 1], self.rewards[t], self.masks[t + 1]) for t in range(T)]

    def step(self, step, action, value_pred=None, action_log_prob=None, state=None):
        """envs.step(action) + the bookkeeping + rollouts.insert(...) of rl/train.py:79-98.
        `action`: [N] or [N, 1] integer tensor on the device."""
        e = self.env
        a = action
        if a.dtype not in (torch.uint8, torch.int32, torch.int64):
            a = a.long()
        if not a.is_contiguous():
            a = a.contiguous()
        if a.numel() != e.num_envs or a.device != e.device:
            raise ValueError("action must hold %d elements on %s" % (e.num_envs, e.device))
        if self._ptr is None:
            self._pointers()
        P = self._ptr
        stream = self._stream()
        ap, at = C.c_void_p(a.data_ptr()), a.element_size()
        e._before_step(a)
        if self.norm is not None:
            z = self.norm
            if not (z.ob and z.ret_on):
                raise NotImplementedError("DeviceRollout drives VecNormalize(ob=True, ret=True), the trainer's configuration")
            _lib.check(self._L.sf_step_normalize(e._h, z._h, ap, at, P["obs"][step + 1], P["r"], P["d"], P["i"],
                                                 C.c_void_p(z._rew.data_ptr()), 0 if z.training else 1, stream))
            _lib.check(self._L.sf_record_step_f32(e.num_envs, C.c_void_p(z._rew.data_ptr()), P["d"], P["rew"][step],
                                                  P["mask"][step + 1], P["ep"], P["fin"], ap, at, P["act"][step], stream))
        else:
            self._step_record(ap, at, step, stream)
        e._stepped(a, self._rew, self._done, self._info)  # (the engine's own int reward, whatever VecNormalize makes of it)
        if value_pred is not None:
            self.value_preds[step].copy_(value_pred)
        if action_log_prob is not None:
            self.action_log_probs[step].copy_(action_log_prob)
        if state is not None:
            self.states[step + 1].copy_(state)
        return self._views[step]

    def _step_record(self, ap, at, step, stream):
        e, P = self.env, self._ptr
        if self.num_stack > 1:
            # the step without an observation, then ONE render launch builds observations[step + 1] from
            # observations[step]: shift by a frame, zero the finished envs, new frame last
            _lib.check(self._L.sf_step_record(e._h, ap, at, None, P["r"], P["d"], P["i"], P["rew"][step],
                                              P["mask"][step + 1], P["ep"], P["fin"], P["act"][step], stream))
            if e.default_geometry:
                _lib.check(self._L.sf_render_shift(e._h, P["obs"][step], P["obs"][step + 1], self.num_stack, P["d"], stream))
            else:
                # the general renderer (another surface geometry) draws into ONE slot: the shift is a copy here, then the new
                # frame into the last slot with the finished envs' older slots zeroed by the same launch (sf_render_stack)
                self.observations[step + 1][:, :-1].copy_(self.observations[step][:, 1:])
                _lib.check(self._L.sf_render_stack(e._h, P["obs"][step + 1], self.num_stack, self.num_stack - 1, P["d"], stream))
            return
        # one launch: the step kernel's epilogue does the trainer's bookkeeping (sfmi.h: sf_step_record)
        _lib.check(self._L.sf_step_record(e._h, ap, at, P["obs"][step + 1], P["r"], P["d"], P["i"], P["rew"][step],
                                          P["mask"][step + 1], P["ep"], P["fin"], P["act"][step], stream))

    @property
    def num_destruction(self):
        """num_destruction += sum(info) of rl/train.py:81, from the step kernel's own accumulator (synchronises)."""
        return int(self.env.episode_stats()[3]) - self._kills0

    def compute_returns(self, next_value, use_gae, gamma, tau):
        """rl/storage.py:50-63 in one launch."""
        n = self.env.num_envs
        nv = next_value.reshape(n).float().contiguous()
        _lib.check(self._L.sf_compute_returns(self.num_steps, n, _p(self.rewards), _p(self.value_preds), _p(self.masks),
                                              _p(nv), _p(self.returns), int(bool(use_gae)), float(gamma), float(tau),
                                              self._stream()))

    def after_update(self):
        """rl/storage.py:45-48."""
        self.observations[0].copy_(self.observations[-1])
        self.states[0].copy_(self.states[-1])
        self.masks[0].copy_(self.masks[-1])

    # ------------------------------------------------------------------ PPO sampling (rl/storage.py:66-122)
    def feed_forward_generator(self, advantages, num_mini_batch):
        """Random minibatches over the T x N transitions; same tuple and shapes as the reference's generator."""
        T, n = self.rewards.shape[0:2]
        batch = T * n
        assert batch >= num_mini_batch, "ppo req batch size to be greater than number of mini batches"
        mb = batch // num_mini_batch
        perm = torch.randperm(batch, device=self.env.device)
        obs = self.observations[:-1].reshape(batch, *self.observations.shape[2:])
        flat = lambda t: t.reshape(batch, t.shape[-1])
        states, actions, returns, masks = flat(self.states[:-1]), flat(self.actions), flat(self.returns[:-1]), flat(self.masks[:-1])
        logp, adv = flat(self.action_log_probs), advantages.reshape(batch, 1)
        for s in range(0, batch, mb):  # BatchSampler(..., drop_last=False)
            idx = perm[s:s + mb]
            yield obs[idx], states[idx], actions[idx], returns[idx], masks[idx], logp[idx], adv[idx]

    def recurrent_generator(self, advantages, num_mini_batch):
        """Whole trajectories of num_processes // num_mini_batch random envs per minibatch, concatenated env by env."""
        n = self.rewards.shape[1]
        per = n // num_mini_batch
        perm = torch.randperm(n, device=self.env.device)
        cat = lambda t, idx: t[:, idx].transpose(0, 1).reshape(-1, *t.shape[2:])
        for s in range(0, n, per):
            idx = perm[s:s + per]
            yield (cat(self.observations[:-1], idx), cat(self.states[:-1], idx), cat(self.actions, idx),
                   cat(self.returns[:-1], idx), cat(self.masks[:-1], idx), cat(self.action_log_probs, idx),
                   cat(advantages, idx))
