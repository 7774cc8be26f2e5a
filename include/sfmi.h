/*
 * sfmi.h -- C ABI of libsfmi.so, the MI355X-native batched Space Fortress engine.
 *
 * This is the drop-in boundary for the env.step() hot path.  The reference binds
 * its engine to Python one environment at a time through the CPython type
 * `_spacefortress.Game` (SRC/pymodule.cpp:361-411, SRC = python/spacefortress/src
 * of the reference) and drives N of them from N processes
 * (gym_vecenv.SubprocVecEnv, rl/train.py:30-32).  libsfmi.so replaces that whole
 * stack -- Game + SSF_Env.step + the vec-env worker loop -- with one batch handle
 * whose state lives in HBM; each entry point below names the reference interface
 * it stands in for.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch / HIP types in signatures;
 *     `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - every *_dev pointer is DEVICE memory owned by the caller (PyTorch); the
 *     library borrows it for the duration of the stream work it enqueues and
 *     makes no allocation after sf_create (one exception: the first image frame
 *     of a batch allocates that batch's render caches).
 *   - calls are asynchronous and stream-ordered; nothing synchronises unless
 *     documented (sf_get_field / sf_episode_stats do).
 *   - every function returns an sf_status (0 = ok, < 0 = error);
 *     sf_last_error() gives the text for the calling thread.
 *   - there is NO CPU fallback: without a usable HIP device sf_create fails with
 *     SF_ERR_NO_DEVICE / SF_ERR_HIP.
 */
#ifndef SFMI_H
#define SFMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFMI_VERSION 1

/* SRC/game.hh:3-4, exported by the reference module at SRC/pymodule.cpp:472-473 */
#define SF_MAX_MISSILES 20
#define SF_MAX_SHELLS 20
/* SRC/game.hh:15-17, exported at SRC/pymodule.cpp:467-470; bit (key-1) of a key mask */
#define SF_FIRE_KEY 1
#define SF_THRUST_KEY 2
#define SF_LEFT_KEY 3
#define SF_RIGHT_KEY 4

typedef enum {
  SF_OK = 0,
  SF_ERR_PRESET = -1,    /* unknown gametype: reference raises RuntimeError (SRC/pymodule.cpp:341) */
  SF_ERR_ARG = -2,       /* bad argument: reference raises from PyArg_ParseTuple / assert (ENV:51) */
  SF_ERR_HIP = -3,       /* a HIP call failed */
  SF_ERR_NO_DEVICE = -4, /* no GPU visible: the product path never falls back to the CPU */
  SF_ERR_ACTION = -5,    /* an action index was out of range (reference: IndexError/KeyError, ENV:211-212) */
  SF_ERR_FIELD = -6,     /* unknown field id / size mismatch in sf_get_field / sf_set_field */
  SF_ERR_STATE = -7      /* sf_check_state: a per-episode counter or timer outgrew its packed width (the reference keeps
                            plain ints, SRC/game.hh:29-43; only batches with SF_FLAG_NO_AUTO_RESET can get there) */
} sf_status;

/* obs_type of SSF_Env (ENV:50-52) */
#define SF_OBS_FEATURES 0   /* ENV:134-157: 19 (youturn) / 17 (autoturn) values */
#define SF_OBS_NORMALIZED 1 /* ENV:109-133 */
#define SF_OBS_MONITORS 2   /* ENV:96-108: 10 values */
#define SF_OBS_NONE 3       /* skip the observation epilogue */
#define SF_OBS_IMAGE 4      /* ENV:203-206 + rl/envs.py:28-30: uint8 [1][84][84], what the trainer's VecEnv yields */
#define SF_OBS_IMAGE_RAW 5  /* ENV:203-206 alone: uint8 [92][90], SSF_Env's own `game_state` */

/* image geometry (ENV:57-58, rl/envs.py:29) */
#define SF_IMAGE_W 90
#define SF_IMAGE_H 92
#define SF_IMAGE_OUT 84

/* flags */
#define SF_FLAG_OBS_F64 1u          /* write observations as float64 (the reference's dtype) instead of float32 */
#define SF_FLAG_REAL_SHELL_COUNT 2u /* feature 14 = live shells; default reproduces the reference, whose
                                       `shells` getter walks the missiles (SRC/pymodule.cpp:131-134) */
#define SF_FLAG_NO_AUTO_RESET 4u    /* bare SSF_Env semantics: a finished lane keeps ticking until sf_reset
                                       (the default is the vec-env worker's reset-on-done, rl/train.py:80) */

#define SF_FLAG_REF_RESET_OBS 8u    /* the observation of a NEW game (sf_reset, a finished lane's auto-reset) carries aim = vdir =
                                       ndist = 0: what the reference's wrapper returns there on fresh memory -- Game::Game leaves
                                       mExtra unwritten (SRC/game.cpp:78) and reset() reads it before the first tick (ENV:163-178).
                                       Default: computeExtra(spawn state), a defined value.
                                       Pinned by tests/golden/wrapper (the real ssf_env.py, executed). */

/* element type of the action array handed to sf_step */
#define SF_ACT_U8 1
#define SF_ACT_I32 4
#define SF_ACT_I64 8

typedef struct {
  const char* gametype;    /* "youturn" | "autoturn" | "test-youturn" | "test-autoturn" (SRC/pymodule.cpp:332-339) */
  int32_t n_envs;          /* environments in this batch (lanes) */
  int32_t device_id;       /* HIP device ordinal */
  int32_t action_set;      /* ENV:50,67-89: 1 = reduced (5 / 3 actions), 0 or -1 = every key combination */
  int32_t obs_type;        /* SF_OBS_* */
  uint32_t flags;          /* SF_FLAG_* */
  uint32_t seed;           /* libc rand() seed of every env process; the reference never seeds => 1 */
  int32_t spawn_skip;      /* spawns already drawn from the stream before lane 0's first Game
                              (rl/train.py:17 draws one in the parent before forking) */
  int32_t spawn_stride;    /* extra skip per lane; 0 = every lane sees the same stream, as in the reference */
  int32_t spawn_table_len; /* entries of the precomputed spawn sequence (power of two <= 2^24).  0 = the smallest power
                              of two >= 65536 that reaches SF_SPAWN_MARGIN entries past the LAST lane's start
                              (spawn_skip + spawn_stride * (n_envs - 1)), so that every lane continues the libc stream
                              from its own offset for at least that many respawns (about 370 episodes); a lane whose
                              cursor passes the end wraps to entry 0.  A table that ends before the last lane's start
                              is SF_ERR_ARG. */
} sf_create_params;

typedef struct sf_batch sf_batch; /* opaque: device state + constant tables */

/* ---- lifecycle: `sf.Game(...)` x N + SSF_Env.__init__ (SRC/pymodule.cpp:319-354, ENV:50-93) ---- */
int sf_create(const sf_create_params* params, sf_batch** out);
int sf_destroy(sf_batch* b); /* tp_dealloc, SRC/pymodule.cpp:295-302 */

int sf_n_envs(const sf_batch* b);
int sf_obs_dim(const sf_batch* b);   /* shape[-1] of the observation (ENV:175) */
int sf_n_actions(const sf_batch* b); /* action_space.n (ENV:90) */
int sf_tick_ms(const sf_batch* b);   /* ENV:61 -> 34 */
int sf_max_ticks(const sf_batch* b); /* ENV:165 -> 5294 */

/* ---- VecEnv.reset(): env.reset() in every worker (ENV:163-178; a brand-new Game per lane,
 *      prev_vlner kept).  obs_dev: [n_envs, obs_dim] f32 (f64 with SF_FLAG_OBS_F64), may be NULL. ---- */
int sf_reset(sf_batch* b, void* obs_dev, void* stream);

/* ---- VecEnv.step(actions): SSF_Env.step in every worker (ENV:208-253 -> press_key/release_key x4|x2
 *      + step_one_tick(34) + is_game_over, SRC/pymodule.cpp:199-240 -> Game::stepOneTick
 *      SRC/game.cpp:473-485), reward shaping (ENV:233-244), feature vector (ENV:95-157) and the
 *      worker's auto-reset (rl/train.py:80).  One fused kernel launch.
 *      actions_dev [n_envs] of act_type; obs_dev [n_envs, obs_dim]; reward_dev int32[n_envs];
 *      done_dev, info_dev uint8[n_envs] (info is the bare fort_kill bool, ENV:253).
 *      Any output pointer may be NULL.  Every env plays the game it would play alone, whatever the batch's size and
 *      however the launch is shaped for it (workgroups of 64 / 128 / 256 envs, a second wave per tile for the missile pool
 *      up to 65 536 envs, several waves per SIMD beyond): checked between 1 and 1 048 576 envs per batch. ---- */
int sf_step(sf_batch* b, const void* actions_dev, int act_type, void* obs_dev, int32_t* reward_dev,
            uint8_t* done_dev, uint8_t* info_dev, void* stream);

/* ---- VecEnv.step on actions the lanes draw themselves, inside the same launch: what the random-action rollout of
 *      BASELINE.json does with `torch.randint` per step, and what stands where the reference's loop has the policy's
 *      sample (rl/train.py:76-80), without an action array to generate, store and load.  Lane l of tile t plays
 *      floor(x * n_actions / 2^32), x = the first word of Philox4x32-10 with key = the 64-bit seed and counter
 *      (first_lane + 64 t + l, tick, 0, 0); `tick` counts this batch's sampled steps since sf_seed_actions (kept per
 *      tile on the device as a uint32, so the call is a pure stream operation: capturable in a HIP graph, every replay
 *      draws new actions; after 2^32 sampled steps -- five days at a microsecond each -- the counter wraps and the stream of
 *      actions repeats).  actions_out_dev, uint8 [n_envs] (may be NULL), receives what was played -- replaying those through
 *      sf_step from the same state gives the same results (tests/test_gpu_sampled.py).  sf_create seeds with
 *      (params.seed, first_lane 0); sf_seed_actions restarts the sequence at tick 0 (synchronises `stream`).
 *      sf_rollout_sampled: the fused n_steps form, actions_out_dev uint8 [n_steps][n_envs]. ---- */
int sf_seed_actions(sf_batch* b, uint64_t seed, uint32_t first_lane, void* stream);
int sf_step_sampled(sf_batch* b, uint8_t* actions_out_dev, void* obs_dev, int32_t* reward_dev, uint8_t* done_dev,
                    uint8_t* info_dev, void* stream);
int sf_rollout_sampled(sf_batch* b, int n_steps, uint8_t* actions_out_dev, void* obs_dev, int32_t* reward_dev,
                       uint8_t* done_dev, uint8_t* info_dev, void* stream);

/* ---- K VecEnv.step calls whose actions are all known up front (open-loop rollouts: random-action
 *      benchmarks, replaying recorded action sequences, evaluating fixed plans), fused into ONE
 *      launch: each wave keeps its environments in registers between ticks, so a tick costs
 *      neither the state round trips through HBM nor a kernel boundary.  Bit-identical to n_steps
 *      consecutive sf_step calls.  actions_dev [n_steps][n_envs]; obs_dev [n_steps][n_envs][obs_dim];
 *      reward_dev / done_dev / info_dev [n_steps][n_envs]; any output may be NULL.
 *      n_steps * n_envs * 8 must stay below 2^32. ---- */
int sf_rollout(sf_batch* b, const void* actions_dev, int act_type, int n_steps, void* obs_dev,
               int32_t* reward_dev, uint8_t* done_dev, uint8_t* info_dev, void* stream);

/* ---- Game.draw() + pb_pixels -> grey frame (SRC/pymodule.cpp:243-254, SRC/draw.cpp:257-270,
 *      ENV:203-206), optionally followed by the trainer's 84x84 INTER_AREA shrink (rl/envs.py:28-30),
 *      for the CURRENT state of every env: what `render()` / `_draw()` give in the reference,
 *      whatever obs_type the batch was created with.  mode SF_OBS_IMAGE: frames_dev uint8
 *      [n_envs][84][84]; SF_OBS_IMAGE_RAW: uint8 [n_envs][92][90].  frames_dev must be 16-byte
 *      aligned.  env_stride = bytes from one env's frame to the next (0 = dense); a larger
 *      stride writes straight into one slot of a [n_envs][num_stack][84][84] frame stack
 *      (rl/train.py:39,51-56), multiple of 16 (raw: 8).  With obs_type SF_OBS_IMAGE / SF_OBS_IMAGE_RAW, sf_reset and sf_step write these
 *      frames to obs_dev themselves (sf_obs_dim = 7056 / 8280 bytes per env); sf_rollout with
 *      obs_dev [n_steps][n_envs][frame] then issues its ticks as n_steps step launches, each followed
 *      by its frames (the fused launch keeps the state in registers), with obs_dev = NULL it stays fused;
 *      sf_rollout_sampled likewise.  A batch with another geometry (sf_set_image_geometry) asks less of the buffers: 84x84
 *      frames 4-byte aligned with env_stride a multiple of 4, raw frames any alignment.  The pixels are the reference's:
 *      cairo 1.16's rasterisation of SRC/draw.cpp's paths and -- in the default geometry, or with sf_set_score_glyphs -- its
 *      score text, bit for bit on all rows (DESIGN.md "image observation"; tests/golden/frames holds frames drawn by the
 *      reference's own renderer). ---- */
int sf_render(sf_batch* b, int mode, uint8_t* frames_dev, size_t env_stride, void* stream);

/* ---- SSF_Env(scale, viewport, ls) (ENV:50-60 -> sf.Game(width = int(vw * scale), height = int(vh * scale), viewport, lw),
 *      SRC/pymodule.cpp:319-354, SRC/draw.cpp:256-270): the geometry of this batch's frames.  The default -- scale .2,
 *      viewport (130, 80, 450, 460), line width 3: every registered gym id and the trainer use it -- has the fast frame
 *      kernel (92x90 surface, caches, draw records); any other geometry switches the batch to the general renderer (one
 *      workgroup per env, every stroke in place): surface width, height in [84, 251] with width * height <= 49 152, i.e.
 *      cv2's INTER_AREA shrink to 84x84 stays below threefold, and at most 0.75 pixels per user unit either way (beyond
 *      that cairo cuts the explosion's circle into more Bezier segments than the renderer's form of it has).  SF_OBS_IMAGE_RAW frames are then uint8 [h][w]
 *      (sf_image_size; sf_obs_dim follows), sf_render_shift is not available, sf_render_stack clears a finished env's
 *      slots with a launch of its own.  May be called any time between frames; synchronous.  SF_ERR_ARG for a geometry
 *      outside those bounds (the batch keeps the one it had). ---- */
int sf_set_image_geometry(sf_batch* b, double scale, double vp_x, double vp_y, double vp_w, double vp_h, double line_width);
int sf_image_size(const sf_batch* b, int32_t* width, int32_t* height); /* of SF_OBS_IMAGE_RAW frames: 90 x 92 by default */
int sf_image_geometry_is_default(const sf_batch* b); /* 1: the fast frame kernel draws this batch; 0: the general renderer */

/* ---- The score text (drawScore / centeredText, SRC/draw.cpp:147-173: cairo's toy font API, "monospace" bold, 30 user
 *      units).  On cairo's image backend the text is FreeType's 8-bit coverage bitmap of every glyph, blitted at a
 *      whole-pixel origin and composited as grey .5 OVER the frame: data + a placement rule, not a rasteriser.  A batch
 *      draws it from a glyph atlas:
 *          alpha[c][gh][gw]   coverage of '0'..'9' (c = 0..9) and '-' (c = 10), one gw x gh box each (<= 24 x 24)
 *          advance, y0        pixels between consecutive boxes; top row of the boxes
 *          x0[first][last]    left column of the first box, by the string's first ('0'..'9', '-') and last character
 *                             (centeredText centres on the ink width, which depends on them in some geometries)
 *      Default geometry: the BUILT-IN atlas (sf_default_score_glyphs) = 6-pixel DejaVu Sans Mono Bold as this image's
 *      cairo 1.16 + FreeType draw it, equal to the reference's own frames (tests/golden/frames/scores.npz).  Any other
 *      geometry starts WITHOUT an atlas -- the seven-segment fallback, which equals no reference pixels -- until one is set
 *      (tests/golden/frames/make_score_golden.py shows how an atlas is taken from a box's cairo).  sf_set_score_glyphs:
 *      layout + alpha = that atlas for the batch's CURRENT geometry (for the default geometry its ink must stay inside
 *      columns 31..58, rows 1..5); layout NULL = the seven-segment fallback, by name.  sf_set_image_geometry resets to the
 *      geometry's default (built-in / none).  Synchronous; rebuilds the cached pictures. ---- */
typedef struct sf_score_glyphs {
  int32_t gw, gh, advance, y0;
  int16_t x0[11][10];
} sf_score_glyphs;
int sf_set_score_glyphs(sf_batch* b, const sf_score_glyphs* layout, const uint8_t* alpha);
/* the atlas the batch draws with now: *has_atlas = 0 -> the fallback (layout, alpha untouched); alpha_bytes >= 11 gw gh */
int sf_get_score_glyphs(const sf_batch* b, int32_t* has_atlas, sf_score_glyphs* layout, uint8_t* alpha, size_t alpha_bytes);
/* the built-in atlas of the default geometry (host only, no GPU needed): alpha uint8[11][4][4] */
int sf_default_score_glyphs(sf_score_glyphs* layout, uint8_t* alpha, size_t alpha_bytes);

/* One step of the trainer's frame stack in one launch (rl/train.py:51-56,92-97): the 84x84 frame of every env
 * goes to slot `slot` of stack_dev uint8 [n_envs][num_stack][84][84] (16-byte aligned), and an env whose
 * done_dev flag is set (may be NULL) first gets its other slots zeroed. */
int sf_render_stack(sf_batch* b, uint8_t* stack_dev, int num_stack, int slot, const uint8_t* done_dev, void* stream);

/* The same update from one buffer into ANOTHER (the trainer stores the stacked observation of every step,
 * rollouts.observations[step + 1], rl/train.py:98): frames 1 .. num_stack-1 of prev_stack_dev become frames
 * 0 .. num_stack-2 of stack_dev (zeros for an env whose done flag is set), the new frame goes last. */
int sf_render_shift(sf_batch* b, const uint8_t* prev_stack_dev, uint8_t* stack_dev, int num_stack, const uint8_t* done_dev,
                    void* stream);

/* `current_obs *= masks` of the trainer's frame stack (rl/train.py:92-93): zero the bytes_per_env bytes of
 * every env whose done flag is set, touching nothing else.  stack_dev uint8 [n_envs][bytes_per_env]. */
int sf_frame_stack_clear(uint8_t* stack_dev, size_t bytes_per_env, const uint8_t* done_dev, int n_envs, void* stream);

/* ---- telemetry: what happened in a tick, per env, as a bitmask -- the reference's per-tick event strings
 *      (Game::addEvent, SRC/game.cpp:124-127 and its call sites :155,169,186,202,223,342,348,363,371,381,385,
 *      393,417) without their multiplicity and order.  Key bits are STATE changes (the reference logs every
 *      press/release call, ENV:213-229 makes one per key and tick).  Once a buffer is set, every sf_step
 *      writes uint32 [n_envs] and every sf_rollout uint32 [n_steps][n_envs] to it; NULL switches it off. ---- */
#define SF_EV_PRESS_FIRE 0x1u
#define SF_EV_PRESS_THRUST 0x2u
#define SF_EV_PRESS_LEFT 0x4u
#define SF_EV_PRESS_RIGHT 0x8u
#define SF_EV_RELEASE_FIRE 0x10u
#define SF_EV_RELEASE_THRUST 0x20u
#define SF_EV_RELEASE_LEFT 0x40u
#define SF_EV_RELEASE_RIGHT 0x80u
#define SF_EV_MISSILE_FIRED 0x100u      /* "missile-fired" */
#define SF_EV_SHIP_RESPAWN 0x200u       /* "ship-respawn" */
#define SF_EV_EXPLODE_BIGHEX 0x400u     /* "explode-bighex" */
#define SF_EV_EXPLODE_SMALLHEX 0x800u   /* "explode-smallhex" */
#define SF_EV_FORTRESS_RESPAWN 0x1000u  /* "fortress-respawn" */
#define SF_EV_FORTRESS_FIRED 0x2000u    /* "fortress-fired" */
#define SF_EV_SHELL_HIT_SHIP 0x4000u    /* "shell-hit-ship" */
#define SF_EV_HIT_FORTRESS 0x8000u      /* "hit-fortress" */
#define SF_EV_VLNER_INCREASED 0x10000u  /* "vlner-increased" */
#define SF_EV_FORTRESS_DESTROYED 0x20000u /* "fortress-destroyed" */
#define SF_EV_VLNER_RESET 0x40000u      /* "vlner-reset" */
#define SF_EV_HIT_DEAD_FORTRESS 0x80000u /* "hit-dead-fortress" */
#define SF_EV_MISSILE_LEFT 0x100000u    /* a missile left the game area (Stats.missedShots; no string in the reference) */
#define SF_EV_GAME_OVER 0x200000u       /* is_game_over() became true (the env is auto-reset unless SF_FLAG_NO_AUTO_RESET) */
int sf_set_event_output(sf_batch* b, uint32_t* events_dev);

/* Out-of-range actions are executed as NOOP and counted on the device; this reads and clears the
 * count (synchronises `stream`).  Returns SF_ERR_ACTION if any were seen since the last call. */
int sf_check_actions(sf_batch* b, void* stream);

/* The per-episode statistics and the key timers live in bit fields sized for one episode (5 295 ticks).  A batch created
 * with SF_FLAG_NO_AUTO_RESET keeps ticking past game over like the bare SSF_Env (ENV:246) until sf_reset; if it is stepped
 * for several episodes' worth of ticks a field can outgrow its bits (deaths / kills: 255 per game; resets, misses, key
 * presses: 65 535; vlner: 4 095; a key timer: +-32 767 ticks without an edge; time: 2^24 ms).  Every time a field of
 * some env leaves its range (the tick on which it wraps) is counted on the device; this reads the count (synchronises
 * `stream`) and returns SF_ERR_STATE while it is not zero -- the values sf_get_field returns for those envs have wrapped,
 * and stay wrapped: the count is STICKY until sf_reset starts new games everywhere, or until a caller that has rewritten EVERY
 * packed field of every env (a restored checkpoint: SFVecEnv.load_state_dict) says so with sf_clear_state_errors -- a
 * sf_set_field of one field repairs that field and leaves the count alone (other fields, other envs may have wrapped).
 * Auto-resetting batches start every field over at each episode end and cannot get there.  sf_set_field refuses
 * (SF_ERR_ARG) values that do not fit a field, and a `stats` row 3 (ship deaths) that is not the sum of rows 0-2.
 * The hand-over counter of split launches (an internal error, never seen) is reported the same way and by sf_episode_stats
 * (which then still delivers the statistics and clears nothing); it is cleared by nothing: make a new batch. */
int sf_check_state(sf_batch* b, void* stream);
int sf_clear_state_errors(sf_batch* b);

/* ---- state access: the 37 read-only attributes of `Game` (SRC/pymodule.cpp:372-411) in batched
 *      form, plus writes for checkpoint/restore and constructed test states.  `host` is HOST memory,
 *      n_envs * sf_field_info().bytes_per_env bytes, slot-major for per-projectile fields
 *      ([slot][env]).  Synchronous. ---- */
typedef struct {
  const char* name;
  int32_t elem_size;     /* 1, 2, 4 or 8 */
  int32_t count;         /* elements per env (20 for per-slot fields, 13 for stats) */
  int32_t is_float;      /* 1 = IEEE float of elem_size, 0 = integer */
} sf_field_desc;

int sf_n_fields(void);
int sf_field_info(int field_id, sf_field_desc* out);
int sf_field_id(const char* name); /* < 0 if unknown */
int sf_get_field(sf_batch* b, int field_id, void* host, size_t bytes);
int sf_set_field(sf_batch* b, int field_id, const void* host, size_t bytes);
/* The same read into DEVICE memory ([count][n_envs] of the field's element type), ordered on `stream`, no synchronise and no
 * PCIe copy: for on-device bookkeeping between two steps (the batch-level counterpart of the reference's per-Game getters,
 * SRC/pymodule.cpp:24-48,78-105; spacefortress_amd/durations.py keeps SRC/game.hh:98-101's four vectors with it).  Every
 * field but missile_x / missile_y / missile_angle (SF_ERR_FIELD: their per-slot view is made on the host's demand). */
int sf_get_field_dev(sf_batch* b, int field_id, void* dev, size_t bytes, void* stream);

/* ---- episode statistics (host side of rl/train.py:81,84-88,161-164): accumulated on the device
 *      at every episode end; this copies them out (synchronises `stream`) and optionally clears.
 *      out[0]=episodes, [1]=sum of returns, [2]=sum of squared returns, [3]=fortress kills seen by
 *      the wrapper (sum of info), [4]=ship deaths, [5]=shots, [6]=min return, [7]=max return
 *      ([6],[7] are INT64_MAX / INT64_MIN while no episode has ended).  Returns SF_ERR_STATE (out[] filled all the same)
 *      if a split step launch ever gave up on a hand-over (sf_check_state's internal-error case): whoever reads the
 *      statistics learns of it without a second call. ---- */
#define SF_EPISODE_STATS_LEN 8
#define SF_SPAWN_MARGIN 65536 /* default spawn table: entries past the last lane's start (sf_create_params.spawn_table_len) */
int sf_episode_stats(sf_batch* b, int64_t* out, int clear, void* stream);

/* ---- diagnostics: reads (and writes linearly) *bytes_moved bytes of state with the step kernel's
 *      own access pattern -- 16 B per lane, 64-lane rows -- so that rocprofv3's FETCH_SIZE /
 *      WRITE_SIZE counters can be calibrated on a known byte count (tools/pmc_report.py).
 *      which: 0 or 1 (two different 16-byte groups).  Synchronous. ---- */
int sf_calibration_copy(sf_batch* b, int which, size_t* bytes_moved);

/* ---- gym_vecenv.VecNormalize(envs) (rl/train.py:35-36: applied whenever the observation is 1-D): running
 *      mean / variance of the observations and of the discounted returns over the batch, observations
 *      become clip((obs - mean) / sqrt(var + eps), +-clipob), rewards clip(rew / sqrt(ret_var + eps),
 *      +-cliprew).  gym-vecenv==1.0 (requirements.txt:4) is not in the reference tree: the algorithm is
 *      OpenAI baselines' vec_normalize.py / running_mean_std.py of that vintage (statistics updated with
 *      the batch BEFORE it is normalised; `ret = ret * gamma + rew`, never cleared at episode ends). ---- */
typedef struct {
  int32_t n_envs, obs_dim, device_id;
  int32_t obs_f64;                        /* observations are float64 (SF_FLAG_OBS_F64), else float32 */
  int32_t ob, ret;                        /* VecNormalize(ob=True, ret=True) */
  double clipob, cliprew, gamma, epsilon; /* 10, 10, 0.99, 1e-8 */
} sf_normalizer_params;
typedef struct sf_normalizer sf_normalizer;
int sf_normalizer_create(const sf_normalizer_params* params, sf_normalizer** out);
int sf_normalizer_destroy(sf_normalizer* z);
/* VecNormalize.step_wait (obs + rewards) or .reset (obs only: reward pointers NULL).  obs_dev
 * [n_envs][obs_dim] -> obs_out_dev (may alias); reward_dev int32[n_envs] -> reward_out_dev float[n_envs].
 * frozen != 0: normalise with the statistics as they are (evaluation), update nothing. */
int sf_normalize(sf_normalizer* z, const void* obs_dev, void* obs_out_dev, const int32_t* reward_dev,
                 float* reward_out_dev, int frozen, void* stream);
/* sf_step + sf_normalize with the batch reduction riding on the step kernel (its waves sum the observation
 * tile they have just written): step -> merge -> apply, one launch less than the two calls.  obs_dev is written
 * raw by the step and normalised in place; reward_dev keeps the raw int32 rewards (the trainer's episode
 * bookkeeping uses them), reward_out_dev gets the normalised ones.  The batch must have been created with a
 * 1-D observation type and the normalizer with its n_envs / obs_dim / obs_f64. */
int sf_step_normalize(sf_batch* b, sf_normalizer* z, const void* actions_dev, int act_type, void* obs_dev,
                      int32_t* reward_dev, uint8_t* done_dev, uint8_t* info_dev, float* reward_out_dev, int frozen,
                      void* stream);
/* statistics as host doubles [2*obs_dim + 4]: ob mean[D], ob var[D], ret mean, ret var, ob count, ret
 * count; ret_host (may be NULL): the per-env discounted returns [n_envs].  Synchronise `stream`. */
int sf_normalizer_get_state(sf_normalizer* z, double* host, double* ret_host, void* stream);
int sf_normalizer_set_state(sf_normalizer* z, const double* host, const double* ret_host, void* stream);

/* ---- what the trainer does between two env steps, kept on the device (plain device pointers, launched
 *      on `stream` of the current device) so that nothing crosses PCIe inside a rollout.
 *      sf_record_step: rl/train.py:82-88 -- reward_out = float(reward), mask_out = 1 - done,
 *      episode_rewards += reward; final_rewards = final_rewards * mask + (1 - mask) * episode_rewards;
 *      episode_rewards *= mask; actions_out (int64, `rollouts.actions[step]`) = the step's actions.
 *      Any output / accumulator may be NULL.
 *      sf_compute_returns: RolloutStorage.compute_returns (rl/storage.py:50-63) over rewards [T][n],
 *      value_preds [T+1][n] (row T is overwritten with next_value when use_gae, like the reference),
 *      masks [T+1][n], next_value [n] -> returns [T+1][n]; float32, bit-identical to the reference. ---- */
/* sf_step with sf_record_step's bookkeeping in the SAME launch (the epilogue of the step kernel): one launch per
 * trainer step.  reward_f32 is required; mask_f32, episode_rewards, final_rewards, actions_out may be NULL. */
int sf_step_record(sf_batch* b, const void* actions_dev, int act_type, void* obs_dev, int32_t* reward_dev,
                   uint8_t* done_dev, uint8_t* info_dev, float* reward_f32, float* mask_f32, float* episode_rewards,
                   float* final_rewards, int64_t* actions_out, void* stream);
int sf_record_step(int n, const int32_t* reward_dev, const uint8_t* done_dev, float* reward_out, float* mask_out,
                   float* episode_rewards, float* final_rewards, const void* actions_dev, int act_type,
                   int64_t* actions_out, void* stream);
/* ... on rewards that are already float (the trainer behind VecNormalize sees normalised rewards) */
int sf_record_step_f32(int n, const float* reward_dev, const uint8_t* done_dev, float* reward_out, float* mask_out,
                       float* episode_rewards, float* final_rewards, const void* actions_dev, int act_type,
                       int64_t* actions_out, void* stream);
int sf_compute_returns(int num_steps, int n, const float* rewards, float* value_preds, const float* masks,
                       const float* next_value, float* returns, int use_gae, double gamma, double tau, void* stream);

/* ---- host-only helpers (usable without a GPU) ---- */
typedef struct {
  int32_t width, height, game_time;
  int32_t destroy_fortress, ship_death_penalty;
  double missile_penalty;
  int32_t miss_penalty;
  int32_t shell_speed, shell_radius, missile_speed, missile_radius;
  int32_t auto_turn;
  int32_t sector_size, lock_time, vuln_time, vuln_threshold, fortress_radius;
  int32_t big_hex, small_hex;
  int32_t explode_duration;
  double start_vx, start_vy;
  int32_t ship_radius;
  double ship_accel;
  int32_t turn_speed;
  int32_t shaped;   /* ENV:235: reward shaping applies to "autoturn"/"youturn" only */
  int32_t n_keys;   /* 4 for youturn games, 2 for autoturn games (ENV:213-229) */
} sf_preset;

/* `Game.config(key)` for the keys the path reads (SRC/pymodule.cpp:266-288, SRC/configs.cpp:3-89) */
int sf_preset_get(const char* gametype, sf_preset* out);
/* ENV:64-89: key mask per action index; out_keys has room for 16; returns the count or < 0 */
int sf_action_table(const char* gametype, int action_set, uint8_t* out_keys);
/* accepted spawns (x, y, angle) of resetShip() (SRC/game.cpp:133-149) for libc seed `seed`:
 * out is int16[n][4] (x, y, angle, 0) */
int sf_spawn_table(uint32_t seed, int n, int16_t* out);
/* cos/sin(deg2rad(k)), k = 0..359, as the reference evaluates them (SRC/vector.cpp:34-36 + libm):
 * out is double[360][2] */
int sf_trig_table(double* out);
/* hexagon vertices (SRC/hexagon.cpp:13-34): out is double[6][2] */
int sf_hex_points(int radius, double* out);

/* the part of every frame that never changes: both hexagons stroked on black (SRC/draw.cpp:131-143,
 * 230-231, 262-263) as 8-bit grey: out is uint8[92][90] */
int sf_image_background(uint8_t* out);
/* ... for any geometry (sf_set_image_geometry): a w x h surface under scale(scale) translate(-vx, -vy), line width lw user
 * units; out is uint8[h][w] */
int sf_image_background_geom(int w, int h, double vx, double vy, double vw, double vh, double lw, uint8_t* out);
/* the live fortress's 8-bit coverage over its 16 x 16 box of the default surface, heading 10 * sector (one cairo_stroke of four
 * lines, SRC/draw.cpp:238-242); the explosion's 86 arcs as Bezier constants, 8 doubles each (SRC/draw.cpp:116-145) */
int sf_image_fort_alpha(int sector, uint8_t* out256);
int sf_arc_table(double* out);
/* test hooks: ONE wireframe (kind 0 ship, 1 fortress, 2 missile, 3 shell; SRC/wireframe.cpp:11-67) as 8-bit coverage, and
 * drawExplosion composited onto fb, both through the frame kernels' formulation (sf_tor.h) run on the host */
int sf_image_object_alpha(int kind, double x, double y, int angle_deg, int w, int h, double vp_x, double vp_y, double vp_w,
                          double vp_h, double lw, uint8_t* alpha);
int sf_image_explosion_host(double x, double y, int w, int h, double vp_x, double vp_y, double vp_w, double vp_h, double lw,
                            uint8_t* fb);
int sf_trig_deg(int deg, double* cos_sin);
/* ... and ONE arc cairo_arc(xc, yc, r, a1, a2) + cairo_stroke, a2 - a1 <= pi / 2; returns the number of pieces it was flattened into */
int sf_image_arc_alpha(double xc, double yc, double r, double a1, double a2, int w, int h, double vp_x, double vp_y, double vp_w,
                       double vp_h, double lw, uint8_t* alpha);
/* the background plus the overlays the render kernel starts from when they are static: variant bit 0 =
 * the score text "0000000" (built-in glyph atlas), bit 1 = the vulnerability bar at 0 (drawScore / drawVlner,
 * SRC/draw.cpp:161-173,207-225); out is uint8[92][90] */
int sf_image_static(int variant, uint8_t* out);
/* cv2.resize(..., INTER_AREA) taps for one axis, ssize -> dsize with dsize <= ssize < 3*dsize
 * (rl/envs.py:29: 90 -> 84 and 92 -> 84 in the default geometry): destination i reads source cells first[i] ..
 * first[i] + count[i] - 1 (count <= 4) with weights alpha[4*i ..]; alpha is float[dsize][4] */
int sf_resize_area_tab(int ssize, int dsize, int32_t* first, int32_t* count, float* alpha);
/* cv2.resize(src, (dw, dh), interpolation=INTER_AREA) for 8-bit grey frames and a shrink below 3x per
 * axis (below 3x: sf_resize_area_tab), with OpenCV's float arithmetic and rounding: the library's own resampling of the static
 * background; host memory, row-major */
int sf_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);

/* The launch order of sf_render* (image batches).  The step kernel marks, one 64-bit word per tile of 64 envs, the
 * envs whose ship died in the tick; the next render launch starts those frames first (the first frame of an
 * explosion is the expensive one).  The words only decide WHEN a frame is drawn, never what it holds; this entry
 * overwrites them from host memory (words[ceil(n_envs / 64)]), for tests that check exactly that.  SF_ERR_ARG for a
 * batch without image observations. */
int sf_set_render_order_hint(sf_batch* b, const uint64_t* words_host, int n_words);

/* ---- diagnostics: the envs' DRAW RECORDS (spacefortress_amd/csrc/sf_drawrec.h), SF_DRAW_RECORD_BYTES per env: what the
 *      frame kernel reads instead of the state -- a 32-byte header of finished decisions (which background, which cached
 *      pictures apply) and 22 transforms (x, y, cos, sin as float32: ship, fortress, 20 missile slots).  The step launches
 *      of an image batch leave them behind; any other change of the state marks them stale and the next frame rebuilds them
 *      from the state.  from_state = 0: the records as they are (SF_ERR_ARG if the batch has none yet); 1: rebuilt from the
 *      state into the batch's buffer first.  host: n_envs * SF_DRAW_RECORD_BYTES bytes.  Synchronous.  Tests compare the
 *      two ways of making them byte for byte (live entries). ---- */
#define SF_DRAW_RECORD_BYTES 432
int sf_draw_records(sf_batch* b, void* host, size_t bytes, int from_state);

const char* sf_last_error(void);
int sf_version(void);
/* hash of the sources and compiler flags this binary was built from (spacefortress_amd/build.py: source_hash) */
const char* sf_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* SFMI_H */
