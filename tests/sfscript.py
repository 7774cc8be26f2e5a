"""Scripted play for the parity tests: action sequences that reach what random actions never do (0 fortress
kills in 200 random episodes) -- the vlner state machine, destroy, the fortress's respawn, shell and hexagon deaths."""
import numpy as np

# FIRE every 8 ticks = 272 ms > the 250 ms vulnerability window, eleven times (vlner 0 -> 11), then a double shot inside
# the window: destroy (SRC/game.cpp:353-402).  FIRE is action 1, THRUST action 2 in both reduced action sets (ENV:71-89).
HUNTER_PATTERN = np.array(([1] + [0] * 7) * 11 + [1, 0, 1, 0] + [0] * 4, np.uint8)


def open_loop_actions(policy, shape, n_actions, rng, phase=0):
    """`shape` = T or (T, N).  `hunter`: the firing pattern (per-lane `phase`), 10 % random actions mixed in (they shift the
    rhythm: vlner resets happen too) -- kills in autoturn games, where the ship aims itself; `charger`: THRUST half of the
    time: deaths at both hexagons, shells fired at a ship that comes close, respawns (SRC/game.cpp:133-157,194-216,314-351);
    `random`: uniform."""
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    acts = rng.integers(0, n_actions, shape).astype(np.uint8)
    if policy == "hunter":
        t = np.arange(shape[0]).reshape((-1,) + (1,) * (len(shape) - 1))
        acts = np.where(rng.random(shape) < 0.1, acts, HUNTER_PATTERN[(t + phase) % len(HUNTER_PATTERN)]).astype(np.uint8)
    elif policy == "charger":
        acts = np.where(rng.random(shape) < 0.5, np.uint8(2), acts).astype(np.uint8)
    elif policy != "random":
        raise ValueError(policy)
    return acts


def aimed_hunter_actions(env, T, rng):
    """Closed-loop hunter (in youturn games nothing else aims the ship): played step by step on `env` (an
    oracle.OracleEnv with the reduced action set, features observation).  A shot on every 8th tick while vlner < 11
    (feature 11), shots on alternate ticks once it is 11 (the double shot that destroys, SRC/game.cpp:364-384); in between
    the ship turns towards the fortress while |aim| > 3 degrees (feature 6, SRC/game.cpp:299-305; LEFT = action 3,
    RIGHT = action 4, ENV:71-78); 3 % random actions.  Returns the actions played: replaying them from the same start
    gives the same game."""
    acts = np.empty(T, np.uint8)
    obs = env.features()
    for t in range(T):
        aim, vlner = obs[6], obs[11]
        if rng.random() < 0.03:
            a = int(rng.integers(0, env.n_actions))
        elif vlner >= 11:
            a = 1 if t % 2 == 0 else 0
        elif t % 8 == 0:
            a = 1
        elif env.n_actions >= 5 and obs[0] and abs(aim) > 3:
            a = 4 if aim > 0 else 3
        else:
            a = 0
        acts[t] = a
        obs, _, done, _ = env.step(a)
        if done:
            obs = env.reset()
    return acts
