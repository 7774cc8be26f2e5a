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
#define SF_LDS_DOUBLES 768 /* the whole block is staged: exactly 3 doubles per thread of a 256-block */

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
  double width_d, height_d;  // width/height as the doubles the comparisons promote them to
  double ndist_a, ndist_b;   // small_hex and (big_hex-small_hex)/2.0 of normDist (SRC/game.cpp:282-284)
  // observation
  int obs_type, obs_f64, real_shell_count, obs_dim, auto_reset;
  double pb_width, pb_height, max_ticks; // ENV:57-58,165
  // episode accumulators / error counter (device)
  unsigned long long* acc;   // SF_EPISODE_STATS_LEN + 1 words; [8] = bad-action count
  unsigned long long* dbg;   // SF_STAMPS diagnostic builds only: [wave][16] clock stamps; else null
};
