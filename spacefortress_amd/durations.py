"""Batch-level duration telemetry: the reference's `mThrustDurations`, `mShotDurations`, `mShotIntervalsInvul`,
`mShotIntervalsVul` (SRC/game.hh:98-101; pushed by Game::processKeyState, SRC/game.cpp:237-261; read through the getters
SRC/pymodule.cpp:143-181) for EVERY env of a batch, on the device.

The pushes are pure functions of the key calls of the tick (SSF_Env.step: press or release of FIRE and THRUST, ENV:213-220)
and of what they meet -- the fire / thrust flags, the two timers and the vulnerability BEFORE the tick:
    FIRE pressed while up:      shot_intervals_vul if vulnerability > 10 else shot_intervals_invul  <- |fire timer|
    FIRE released while down:   shot_durations   <- fire timer
    THRUST released while down: thrust_durations <- thrust timer
so the log reads four fields on the device in front of the step launch (sfmi.h: sf_get_field_dev: no synchronise, no PCIe) and
appends behind it.  A Game's vectors live as long as the Game: an env that finishes its episode starts a new Game (the vec-env
worker's reset) with empty vectors.  `spacefortress.core.Game` keeps the same vectors for its one env on the host
(tests/golden/getters holds what the reference's extension returned for them)."""
import ctypes as C

import torch

from . import _lib

NAMES = ("thrust_durations", "shot_durations", "shot_intervals_invul", "shot_intervals_vul")


class DurationLog:
    def __init__(self, env, capacity=512):
        self.env, self.capacity = env, int(capacity)
        n, dev = env.num_envs, env.device
        self.values = {k: torch.zeros((n, self.capacity), dtype=torch.int32, device=dev) for k in NAMES}
        self.counts = {k: torch.zeros(n, dtype=torch.int64, device=dev) for k in NAMES}
        self.dropped = torch.zeros((), dtype=torch.int64, device=dev)  # pushes beyond `capacity` (counted, not stored)
        keys = (C.c_uint8 * 16)()
        na = _lib.lib().sf_action_table(env.gametype.encode(), env._create["action_set"], keys)
        self._keys = torch.tensor([int(keys[i]) for i in range(na)], dtype=torch.int64, device=dev)
        self._lane = torch.arange(n, device=dev)
        self._flags = self._fire_t = self._thrust_t = self._vlner = self._k = None

    def before(self, actions):
        e = self.env
        self._flags = e.get_field_tensor("flags", self._flags)
        self._fire_t = e.get_field_tensor("fire_timer", self._fire_t)
        self._thrust_t = e.get_field_tensor("thrust_timer", self._thrust_t)
        self._vlner = e.get_field_tensor("vlner", self._vlner)
        self._k = self._keys[actions.reshape(-1).long().clamp(0, len(self._keys) - 1)]  # (out of range runs as NOOP: key 0 is NOOP)

    def _push(self, name, mask, value):
        c = self.counts[name]
        ok = mask & (c < self.capacity)
        idx = c.clamp(max=self.capacity - 1)
        v = self.values[name]
        v[self._lane, idx] = torch.where(ok, value.to(torch.int32), v[self._lane, idx])
        self.counts[name] = c + ok.long()
        self.dropped += (mask & ~ok).sum()

    def after(self, done):
        fl = self._flags.long()
        fire_down, thrust_down = (fl & 4) != 0, (fl & 8) != 0          # sf_layout.h: flags bit 2 fire, bit 3 thrust
        fire, thrust = (self._k & 1) != 0, (self._k & 2) != 0          # ENV:213-220: press if the action holds the key, else release
        ft, tt = self._fire_t.long(), self._thrust_t.long()
        vul = self._vlner.long() > 10
        new_shot = fire & ~fire_down
        self._push("shot_intervals_vul", new_shot & vul, ft.abs())     # SRC/game.cpp:240-243
        self._push("shot_intervals_invul", new_shot & ~vul, ft.abs())
        self._push("shot_durations", ~fire & fire_down, ft)            # :258-260
        self._push("thrust_durations", ~thrust & thrust_down, tt)      # :253-255
        if self.env._create["auto_reset"]:
            fin = done.reshape(-1) != 0                                 # a new Game: its vectors start empty (SRC/game.cpp:64-67)
            for k in NAMES:
                self.counts[k] = torch.where(fin, torch.zeros_like(self.counts[k]), self.counts[k])

    def reset(self):
        for k in NAMES:
            self.counts[k].zero_()

    def get(self, name):
        """(values int32 [N, capacity], counts int64 [N]): env i's vector is values[i, :counts[i]]."""
        return self.values[name], self.counts[name]

    def of(self, i, name):
        """Env i's vector as a tuple of ints, as the reference's getter returns it (synchronises)."""
        return tuple(int(v) for v in self.values[name][i, :int(self.counts[name][i])].cpu())
