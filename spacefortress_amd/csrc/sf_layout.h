// sf_layout.h -- world state of a batch in HBM, shared by the kernels and the C-ABI host code.
//
// Struct-of-arrays: every field is one array [count][lanes] (slot-major for the
// per-projectile fields), all carved from ONE allocation.  Field k starts at byte
// offset kOffsetPerLane[k] * lanes, where lanes is the batch size padded to a
// multiple of 256, so every array starts 256-B aligned and lane i of a wave
// reads base + i*elem: fully coalesced.  The reference keeps the same data as one
// 2568-byte `Game` object per env (SRC/game.hh:84-143).
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
}  // namespace sfl

// Constant block staged into LDS by every workgroup (doubles):
//   [0, 720)    cos/sin(deg2rad(k)) interleaved, k = 0..359
//   [720, 744)  big hexagon: per edge (nx, ny, px, py)   SRC/hexagon.cpp:36-48
//   [744, 768)  small hexagon, same
#define SF_LDS_TRIG 0
#define SF_LDS_BIGHEX 720
#define SF_LDS_SMALLHEX 744
#define SF_LDS_DOUBLES 768

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
  // preset (SRC/configs.cpp)
  int width, height, game_time, tick_ms;
  int sector_size, lock_time, vuln_time, vuln_threshold;
  int explode_duration, turn_speed, shaped;
  int missile_speed, shell_speed;
  float missile_penalty, death_penalty, destroy_reward, miss_penalty;
  double ship_accel, start_vx, start_vy;
  double missile_hit_r2, shell_hit_r2; // (r1+r2)^2: see sf_kernels.hip on the sqrt-free test
  double fort_x, fort_y;
  double ndist_a, ndist_b;   // small_hex and (big_hex-small_hex)/2.0 of normDist (SRC/game.cpp:282-284)
  // observation
  int obs_type, obs_f64, real_shell_count, obs_dim, auto_reset;
  double pb_width, pb_height, max_ticks; // ENV:57-58,165
  // episode accumulators / error counter (device)
  unsigned long long* acc;   // SF_EPISODE_STATS_LEN + 1 words; [8] = bad-action count
  unsigned long long* dbg;   // SF_STAMPS diagnostic builds only: [wave][16] clock stamps; else null
};
