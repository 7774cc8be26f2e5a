"""`import spacefortress.core as sf` -- the constants the reference module exports
(SRC/pymodule.cpp:467-473).  The per-env `Game` type is replaced by the batched C ABI
(include/sfmi.h); a `Game`-shaped single-lane view is listed under "next" in DESIGN.md."""
FIRE_KEY, THRUST_KEY, LEFT_KEY, RIGHT_KEY = 1, 2, 3, 4
MAX_MISSILES = 20
MAX_SHELLS = 20
