"""FrameStack -- the trainer's `current_obs` for image observations (rl/train.py:38-43,51-56,92-97),
kept on the device as a ring so that a step writes ONE new 84x84 frame per env and moves nothing.

The reference keeps a float tensor [N, num_stack * 1, 84, 84]; every step it shifts the stack by one
frame (`current_obs[:, :-1] = current_obs[:, 1:]`), zeroes the whole stack of finished envs
(`current_obs *= masks`) and stores the new observation last.  Here sf_render writes the new frame
straight into the ring slot (sfmi.h: env_stride), finished envs are zeroed with one masked fill, and
`stacked()` gives the frames in the reference's order (oldest first) when a consumer wants them so.
"""
import ctypes as C

import torch

from . import _lib


class FrameStack:
    def __init__(self, env, num_stack=4):
        if env.obs_type != "image":
            raise ValueError("FrameStack wraps an SFVecEnv created with obs_type='image'")
        self.env = env
        self.num_stack = int(num_stack)
        n, s = env.num_envs, self.num_stack
        self.ring = torch.zeros((n, s, _lib.IMAGE_OUT, _lib.IMAGE_OUT), dtype=torch.uint8, device=env.device)
        self.head = s - 1  # slot of the newest frame
        self._rew = torch.empty(n, dtype=torch.int32, device=env.device)
        self._done = torch.empty(n, dtype=torch.uint8, device=env.device)
        self._info = torch.empty(n, dtype=torch.uint8, device=env.device)

    def _slot(self, k):
        return self.ring[:, k:k + 1]  # [N, 1, 84, 84] view, env stride = num_stack frames

    def reset(self):
        """envs.reset() + update_current_obs(obs) on a zeroed stack (rl/train.py:43,60-61)."""
        e = self.env
        e._touch()  # (a recording cannot go on across a reset: vecenv.py)
        _lib.check(e._L.sf_reset(e._h, None, e._stream()))
        self.ring.zero_()
        self.head = self.num_stack - 1
        e.render("image", out=self._slot(self.head))
        return self.stacked()

    def step(self, actions):
        """One VecEnv step; returns (reward int32 [N], done bool [N], info bool [N]) -- views of buffers the
        next step overwrites; the stack is updated in place: finished envs keep only the first frame of their next
        episode."""
        e = self.env
        at = {torch.uint8: 1, torch.int32: 4, torch.int64: 8}[actions.dtype]
        e._before_step(actions)
        _lib.check(e._L.sf_step(e._h, C.c_void_p(actions.data_ptr()), at, None, C.c_void_p(self._rew.data_ptr()),
                                C.c_void_p(self._done.data_ptr()), C.c_void_p(self._info.data_ptr()), e._stream()))
        e._stepped(actions, self._rew, self._done, self._info)
        # the new frame into the next slot; finished envs get their other slots zeroed by the same launch
        # (current_obs *= masks, rl/train.py:92-93)
        self.head = (self.head + 1) % self.num_stack
        _lib.check(e._L.sf_render_stack(e._h, C.c_void_p(self.ring.data_ptr()), self.num_stack, self.head,
                                        C.c_void_p(self._done.data_ptr()), e._stream()))
        return self._rew, self._done.view(torch.bool), self._info.view(torch.bool)  # 0 / 1 bytes: a view, not a kernel

    def stacked(self):
        """[N, num_stack, 84, 84] uint8, oldest frame first -- the reference's channel order."""
        return torch.roll(self.ring, -(self.head + 1), dims=1)
