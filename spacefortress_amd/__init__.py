"""spacefortress_amd -- MI355X-native batched Space Fortress env.step().

The compute path is libsfmi.so (hand-written HIP kernels behind the C ABI of
include/sfmi.h); this package is the Python host side that mirrors the
reference's gym / VecEnv surface.  Importing never touches the GPU; creating an
environment does, and fails loudly if the extension or the device is missing.
"""
from ._lib import SfmiError, lib  # noqa: F401

__all__ = ["SFVecEnv", "SSF_Env", "FrameStack", "SFVecNormalize", "DeviceRollout", "Replay", "SfmiError", "lib"]


def __getattr__(name):
    if name == "SFVecEnv":
        from .vecenv import SFVecEnv
        return SFVecEnv
    if name == "SSF_Env":
        from .env import SSF_Env
        return SSF_Env
    if name == "FrameStack":
        from .framestack import FrameStack
        return FrameStack
    if name == "SFVecNormalize":
        from .vecnormalize import SFVecNormalize
        return SFVecNormalize
    if name == "DeviceRollout":
        from .rollout import DeviceRollout
        return DeviceRollout
    if name == "Replay":
        from .replay import Replay
        return Replay
    raise AttributeError(name)
