// sf_layout.h -- world state of a batch in HBM, shared by the kernels and the C-ABI host code.
//
// Struct-of-arrays per WAVE TILE: the batch is cut into tiles of 64 envs (one wavefront);
// a tile is one contiguous block of kTileBytes holding, field after field, the 64 lanes'
// values: [field][slot][64 lanes].  So
//   * lane l of a wave reads  tile_base + C(field, slot) + l*elem : 64 consecutive elements,
//     a fully coalesced row of 64*elem bytes, aligned to its own size;
//   * C(field, slot) is a COMPILE-TIME constant: the whole state of a wave is addressed from
//     one scalar base (an SGPR pair) plus one per-lane byte offset per element size -- no
//     per-field base arithmetic (the first, batch-wide SoA version of this kernel spent a
//     quarter of its instructions on 64-bit address math and SGPR spills);
//   * a wave's working set is one 74 KB block: page- and channel-local.
// The batch is padded to a multiple of 256 envs (4 tiles = one workgroup).  The reference keeps
// the same data as one 2568-byte `Game` object per env (SRC/game.hh:84-143).
#pragma once
#include <stdint.h>

#define SF_NSLOT 20 /* SRC/game.hh:3-4 */
#define SF_NSTAT 13 /* SRC/game.hh:29-43 */

// X(name, ctype, count, is_float) -- ordered by decreasing element size so that every
// array stays naturally aligned.  Reference members in the comments.
#define SF_FIELDS(X)                                                                         \
  X(ship_x, double, 1, 1)         /* mShip.mPos.mX            SRC/object.hh:5 */             \
  X(ship_y, double, 1, 1)         /* mShip.mPos.mY */                                        \
  X(ship_vx, double, 1, 1)        /* mShip.mVel.mX */                                        \
  X(ship_vy, double, 1, 1)        /* mShip.mVel.mY */                                        \
  X(missile_x, double, SF_NSLOT, 1) /* mMissiles[i].mPos      SRC/game.hh:90 */              \
  X(missile_y, double, SF_NSLOT, 1)                                                          \
  X(shell_x, double, SF_NSLOT, 1)  /* mShells[i].mPos, mVel   SRC/game.hh:91 */              \
  X(shell_y, double, SF_NSLOT, 1)                                                            \
  X(shell_vx, double, SF_NSLOT, 1)                                                           \
  X(shell_vy, double, SF_NSLOT, 1)                                                           \
  X(ship_death_timer, int32_t, 1, 0) /* mShip.mDeathTimer     SRC/game.hh:60-64 */           \
  X(fire_timer, int32_t, 1, 0)                                                               \
  X(thrust_timer, int32_t, 1, 0)                                                             \
  X(left_timer, int32_t, 1, 0)                                                               \
  X(right_timer, int32_t, 1, 0)                                                              \
  X(fort_timer, int32_t, 1, 0)       /* mFortress.mTimer      SRC/game.hh:77 */              \
  X(fort_death_timer, int32_t, 1, 0)                                                         \
  X(fort_vuln_timer, int32_t, 1, 0)                                                          \
  X(points, float, 1, 1)             /* mScore                SRC/game.hh:49-52 */           \
  X(raw_points, float, 1, 1)                                                                 \
  X(vlner, int32_t, 1, 0)                                                                    \
  X(time, int32_t, 1, 0)             /* mTime (mTick = mTime / tick_ms) SRC/game.hh:93 */    \
  X(stats, int32_t, SF_NSTAT, 0)     /* mStats                SRC/game.hh:29-43 */           \
  X(prev_vlner, int32_t, 1, 0)       /* SSF_Env.prev_vlner    ENV:92,244 */                  \
  X(spawn_cursor, uint32_t, 1, 0)    /* position in the process's rand() spawn sequence */   \
  X(missile_mask, uint32_t, 1, 0)    /* bit i = mMissiles[i].mAlive */                       \
  X(shell_mask, uint32_t, 1, 0)      /* bit i = mShells[i].mAlive */                         \
  X(ep_return, int32_t, 1, 0)        /* running sum of wrapper rewards (rl/train.py:84) */   \
  X(ep_kills, int32_t, 1, 0)         /* running sum of info (rl/train.py:81) */              \
  X(ship_angle, int16_t, 1, 0)       /* mShip.mAngle: always an integer in [0,360) */        \
  X(fort_angle, int16_t, 1, 0)       /* mFortress.mAngle: multiple of the sector size */     \
  X(fort_last_angle, int16_t, 1, 0)  /* mFortress.mLastAngle */                              \
  X(missile_angle, int16_t, SF_NSLOT, 0) /* mMissiles[i].mAngle (velocity = 20*(cos,sin)) */ \
  X(flags, uint8_t, 1, 0)            /* SF_FL_* bits */

enum SfFieldId {
#define X(name, ctype, count, isf) SF_F_##name,
  SF_FIELDS(X)
#undef X
      SF_F_COUNT
};

// flags bits
#define SF_FL_SHIP_ALIVE 1u
#define SF_FL_FORT_ALIVE 2u
#define SF_FL_FIRE 4u   /* mShip.mFireFlag   SRC/game.hh:65-68 */
#define SF_FL_THRUST 8u /* mShip.mThrustFlag */
#define SF_FL_LEFT 16u  /* mShip.mLeftFlag */
#define SF_FL_RIGHT 32u /* mShip.mRightFlag */

// stats indices (order of SRC/game.hh:29-43 and of the `stats` getter SRC/pymodule.cpp:78-96)
enum {
  SF_ST_BIG_HEX_DEATHS = 0, SF_ST_SMALL_HEX_DEATHS, SF_ST_SHELL_DEATHS, SF_ST_SHIP_DEATHS, SF_ST_RESETS,
  SF_ST_DESTROYED, SF_ST_MISSED, SF_ST_SHOTS, SF_ST_THRUSTS, SF_ST_LEFTS, SF_ST_RIGHTS, SF_ST_VLNER_INCS,
  SF_ST_MAX_VLNER
};

namespace sfl {
struct FieldMeta {
  const char* name;
  int elem_size, count, is_float;
};
constexpr FieldMeta kFields[SF_F_COUNT] = {
#define X(name, ctype, count, isf) {#name, (int)sizeof(ctype), count, isf},
    SF_FIELDS(X)
#undef X
};
// bytes per lane that precede field f
constexpr long offset_per_lane(int f) {
  long o = 0;
  for (int i = 0; i < f; i++) o += (long)kFields[i].elem_size * kFields[i].count;
  return o;
}
constexpr long kBytesPerLane = offset_per_lane(SF_F_COUNT);
constexpr int kTileLanes = 64;                          // one wavefront
constexpr long kTileBytes = kBytesPerLane * kTileLanes;  // 74 432 B
// byte offset, inside a tile, of lane 0 of (field f, slot s)
constexpr long tile_offset(int f, int s = 0) {
  return offset_per_lane(f) * kTileLanes + (long)s * kTileLanes * kFields[f].elem_size;
}
}  // namespace sfl

// Host-built constant block (doubles):
//   [0, 720)    cos/sin(deg2rad(k)) interleaved, k = 0..359 -- indexed per lane, so every
//               workgroup stages it into LDS (SF_LDS_DOUBLES)
//   [720, 744)  big hexagon: per edge (nx, ny, px, py)   SRC/hexagon.cpp:36-48
//   [744, 768)  small hexagon, same -- uniform, so they travel as kernel arguments (SGPRs)
#define SF_LDS_TRIG 0
#define SF_LDS_BIGHEX 720
#define SF_LDS_SMALLHEX 744
#define SF_CONST_DOUBLES 768
#define SF_LDS_DOUBLES 720 /* what a wave stages into LDS: the cos/sin table (indexed per lane) */

// The 12 hexagon edges as Hexagon::isInside forms them (SRC/hexagon.cpp:38-42), X(nx, ny, px, py),
// for radius 200 / 40 (bigHex / smallHex of every preset, SRC/configs.cpp:34-35).  Compiled into the
// kernels as immediates; sf_create checks them bit for bit against what Hexagon::setRadius's
// arithmetic (sf_host.cpp, host libm) gives.  Note the floor-induced asymmetry 174/173, 35/34.
#define SF_BIG_HEX_EDGES(X)                                                                          \
  X(174.0, 100.0, 155.0, 315.0) X(-0.0, 200.0, 255.0, 141.0) X(-174.0, 100.0, 455.0, 141.0)          \
  X(-173.0, -100.0, 555.0, 315.0) X(-0.0, -200.0, 455.0, 488.0) X(173.0, -100.0, 255.0, 488.0)
#define SF_SMALL_HEX_EDGES(X)                                                                        \
  X(35.0, 20.0, 315.0, 315.0) X(-0.0, 40.0, 335.0, 280.0) X(-35.0, 20.0, 375.0, 280.0)               \
  X(-34.0, -20.0, 395.0, 315.0) X(-0.0, -40.0, 375.0, 349.0) X(34.0, -20.0, 335.0, 349.0)

// baseConfig (SRC/configs.cpp:3-49) and the wrapper's constants (ENV:57-61,165): identical in all
// four presets, so they are compile-time constants of the kernels (immediates, not kernel-argument
// loads a lone wave would have to wait for).  sf_create checks every preset against them.
namespace sfc {
constexpr int game_time = 180000, tick_ms = 34, sector_size = 10, lock_time = 1000, vuln_time = 250,
              vuln_threshold = 10, explode_duration = 1000, turn_speed = 6, missile_speed = 20, shell_speed = 6,
              fort_respawn = 1000 /* literal at SRC/game.cpp:199 */;
constexpr double width_d = 710, height_d = 626;         // compared as doubles (SRC/game.cpp:130)
constexpr double fort_x = 355, fort_y = 315;            // SRC/game.cpp:38-39
constexpr double ship_accel = 0.3;
constexpr double missile_hit_r2 = 23.0 * 23.0;          // (missile 5 + fortress 18)^2, see the sqrt-free test
constexpr double shell_hit_r2 = 13.0 * 13.0;            // (shell 3 + ship 10)^2
constexpr double ndist_a = 40, ndist_b = (200 - 40) / 2.0;  // normDist, SRC/game.cpp:282-284
constexpr double pb_width = 90, pb_height = 92, max_ticks = 5294;  // ENV:57-58,165
// the scoring triple: "autoturn"/"youturn" (SRC/configs.cpp:57-59,68-70) vs the test-* presets (:8-10)
template <bool SHAPED>
struct Score {
  static constexpr float missile_penalty = SHAPED ? 0.05f : 2.0f;  // (float)0.05, penalize(float) :104,187
  static constexpr float death_penalty = SHAPED ? 1.0f : 100.0f;   // :339,345,415
  static constexpr float destroy_reward = SHAPED ? 1.0f : 100.0f;  // + mDestroyFortressExtraPoints = 0 (:47,380)
  static constexpr float miss_penalty = 0.0f;                      // :397
};
}  // namespace sfc

// Everything the kernels need that is uniform across lanes; passed by value (kernarg -> SGPRs).
struct SfKernelArgs {
  unsigned char* state;      // base of the SoA block
  long lanes;                // padded lane count (multiple of 256)
  int n_envs;                // real lanes
  const double* consts;      // SF_LDS_DOUBLES doubles in HBM
  const int16_t* spawn;      // [spawn_len][4] (x, y, angle, 0)
  unsigned spawn_mask;       // spawn_len - 1
  unsigned long long action_keys; // 4 bits per action index, up to 16 actions
  int n_actions;
  double start_vx, start_vy; // cos/sin(deg2rad(-60)) as the host's libm gives them (SRC/configs.cpp:43-44)
  // observation
  int obs_type, obs_f64, real_shell_count, obs_dim, auto_reset;
  // episode accumulators / error counter (device)
  unsigned long long* acc;   // SF_EPISODE_STATS_LEN + 1 words; [8] = bad-action count
  unsigned long long* dbg;   // SF_STAMPS diagnostic builds only: [wave][16] clock stamps; else null
};
