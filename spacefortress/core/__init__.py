"""`import spacefortress.core as sf` -- what the reference's extension module exports
(SRC/pymodule.cpp:467-473, 476-509): the key constants, the slot limits, and `Game`, here a single-lane view
of the batched device engine (spacefortress_amd/game.py; the batch itself is include/sfmi.h)."""
from spacefortress_amd.game import FIRE_KEY, THRUST_KEY, LEFT_KEY, RIGHT_KEY, Game  # noqa: F401

MAX_MISSILES = 20
MAX_SHELLS = 20
