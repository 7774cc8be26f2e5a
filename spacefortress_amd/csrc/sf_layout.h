// sf_layout.h -- world state of a batch in HBM, shared by the kernels and the C-ABI host code.
//
// One contiguous block per WAVE TILE: the batch is cut into tiles of 64 envs (one wavefront);
// inside a tile the state is a sequence of GROUPS, each stored [slot][64 lanes][chunk], where a
// chunk is what ONE lane reads or writes with ONE memory instruction:
//   * 16-byte chunks pack the fields that always travel together -- (x, y), (vx, vy), four
//     int32 timers, ... -- so lane l issues one `global_load_dwordx4` at
//     tile_base + C(group, slot) + 16*l and the wave moves 1 KiB of consecutive bytes: the
//     widest, fully coalesced access the hardware has (the guide's 16 B/lane rule);
//   * C(group, slot) is a COMPILE-TIME constant: the whole state of a wave hangs off one
//     scalar base (an SGPR pair) plus one per-lane offset per chunk size;
//   * the per-episode counters (statistics, episode sums) ride in the spare bits of words the lane moves anyway -- an
//     episode is 5 295 ticks, so 8 / 12 / 16 bits hold them (SF_W_* below): no atomics, no extra round trips, no bytes;
//   * live missiles are NOT stored per env and slot: a tile keeps them as one dense POOL of entries
//     (x, y | heading, owner lane, slot), `missile_pos` / `missile_meta` rows of 64 entries, which the wave moves a
//     row at a time, every lane busy, and compacts in place with a ballot + prefix count as entries leave
//     (per env and slot, two lanes in ten had a missile in a slot the wave had to walk anyway);
//   * a wave's working set is one 72 KB block: page- and channel-local.
// History (see HISTORY.md §5): batch-wide [field][N] arrays -> a quarter of the instructions
// were per-field 64-bit address math and SGPR spills; per-tile [field][64] rows -> 24 loads and
// 23 stores of 1-8 bytes per lane and step; this layout -> 7 + 7.
// The batch is padded to a multiple of 256 envs (4 tiles = one workgroup).  The reference keeps
// the same data as one 2568-byte `Game` object per env (SRC/game.hh:84-143).  The layout never
// leaves the library: sf_get_field / sf_set_field gather and scatter by field name.
#pragma once
#include <stdint.h>

#define SF_NSLOT 20 /* SRC/game.hh:3-4 */
#define SF_NSTAT 13 /* SRC/game.hh:29-43 */

// G(group, chunk_bytes, slots)
#define SF_GROUPS(G)                                                                            \
  G(ship_pos, 16, 1)              /* ship_x, ship_y */                                          \
  G(ship_vel, 16, 1)              /* ship_vx, ship_vy */                                        \
  G(timers_a, 16, 1)              /* prev_vlner; fire, thrust, left timers (all first read late in the tick; packed, SF_W_*) */                   \
  G(timers_b, 16, 1)              /* right (packed, SF_W_*), fort, fort_death, fort_vuln timers */               \
  G(score, 16, 1)                 /* points, raw_points, vlner, time (the last two packed, SF_W_*) */                         \
  G(misc, 16, 1)                  /* ship_death timer, spawn_cursor, missile word, shell word (SF_MASK_BITS): what the respawn and the slot allocation need first */ \
  G(small, 16, 1)                 /* ship_angle, fort_angle, fort_last_angle (i16), flags (u8), last_reward (i8); then the four key-press counters of `stats` as u16 (see SF_KEYCOUNT_BYTE) */ \
  G(missile_pos, 16, SF_NSLOT)    /* the tile's missile pool: entry e = row e / 64, lane e % 64: (x, y) */ \
  G(missile_meta, 4, SF_NSLOT)    /* ... and its (heading | owner lane << 9 | slot << 15): SF_MM_* */      \
  G(shell_pos, 16, SF_NSLOT)      /* shell_x, shell_y */                                        \
  G(shell_vel, 16, SF_NSLOT)      /* shell_vx, shell_vy */

enum SfGroupId {
#define G(name, chunk, slots) SF_G_##name,
  SF_GROUPS(G)
#undef G
      SF_G_COUNT
};

// X(name, ctype, count, is_float, group, byte offset inside the lane's chunk, kind)
// `count` = elements per env.  Reference members in the comments.  kind: SF_FK_PLAIN = `count` slots of `group`, one
// element at `byte offset` of the lane's chunk; the other kinds are assembled by sf_launch_field_copy:
//   SF_FK_BITS    a bit field of the 32-bit word at `byte offset` (shift, width, sign: SF_BITFIELDS below)
//   SF_FK_STATS   the 13 counters of SRC/game.hh:29-43 from their bit fields and (ship deaths) their sum
//   SF_FK_EPRET   ep_return, an int32 carried as two 16-bit halves
//   SF_FK_MPOOL   per-slot missile values, gathered from / scattered into the tile's pool by owner and slot
#define SF_FK_PLAIN 0
#define SF_FK_BITS 1
#define SF_FK_STATS 2
#define SF_FK_EPRET 3
#define SF_FK_MPOOL 4
#define SF_FIELDS(X)                                                                                        \
  X(ship_x, double, 1, 1, ship_pos, 0, SF_FK_PLAIN)           /* mShip.mPos.mX            SRC/object.hh:5 */            \
  X(ship_y, double, 1, 1, ship_pos, 8, SF_FK_PLAIN)           /* mShip.mPos.mY */                                       \
  X(ship_vx, double, 1, 1, ship_vel, 0, SF_FK_PLAIN)          /* mShip.mVel.mX */                                       \
  X(ship_vy, double, 1, 1, ship_vel, 8, SF_FK_PLAIN)          /* mShip.mVel.mY */                                       \
  X(missile_x, double, SF_NSLOT, 1, missile_pos, 0, SF_FK_MPOOL) /* mMissiles[i].mPos      SRC/game.hh:90 */            \
  X(missile_y, double, SF_NSLOT, 1, missile_pos, 8, SF_FK_MPOOL)                                                        \
  X(shell_x, double, SF_NSLOT, 1, shell_pos, 0, SF_FK_PLAIN)  /* mShells[i].mPos, mVel    SRC/game.hh:91 */             \
  X(shell_y, double, SF_NSLOT, 1, shell_pos, 8, SF_FK_PLAIN)                                                            \
  X(shell_vx, double, SF_NSLOT, 1, shell_vel, 0, SF_FK_PLAIN)                                                           \
  X(shell_vy, double, SF_NSLOT, 1, shell_vel, 8, SF_FK_PLAIN)                                                           \
  X(ship_death_timer, int32_t, 1, 0, misc, 0, SF_FK_PLAIN) /* mShip.mDeathTimer       SRC/game.hh:60-64 */          \
  X(fire_timer, int32_t, 1, 0, timers_a, 4, SF_FK_BITS)                                                                 \
  X(thrust_timer, int32_t, 1, 0, timers_a, 8, SF_FK_BITS)                                                               \
  X(left_timer, int32_t, 1, 0, timers_a, 12, SF_FK_BITS)                                                                \
  X(right_timer, int32_t, 1, 0, timers_b, 0, SF_FK_BITS)                                                                \
  X(fort_timer, int32_t, 1, 0, timers_b, 4, SF_FK_PLAIN)      /* mFortress.mTimer         SRC/game.hh:77 */             \
  X(fort_death_timer, int32_t, 1, 0, timers_b, 8, SF_FK_PLAIN)                                                          \
  X(fort_vuln_timer, int32_t, 1, 0, timers_b, 12, SF_FK_PLAIN)                                                          \
  X(points, float, 1, 1, score, 0, SF_FK_PLAIN)               /* mScore                   SRC/game.hh:49-52 */          \
  X(raw_points, float, 1, 1, score, 4, SF_FK_PLAIN)                                                                     \
  X(vlner, int32_t, 1, 0, score, 8, SF_FK_BITS)                                                                         \
  X(time, int32_t, 1, 0, score, 12, SF_FK_BITS)               /* mTime (mTick = mTime / tick_ms) SRC/game.hh:93 */      \
  X(stats, int32_t, SF_NSTAT, 0, misc, 0, SF_FK_STATS)       /* mStats                   SRC/game.hh:29-43 */          \
  X(prev_vlner, int32_t, 1, 0, timers_a, 0, SF_FK_BITS)        /* SSF_Env.prev_vlner       ENV:92,244 */                 \
  X(spawn_cursor, uint32_t, 1, 0, misc, 4, SF_FK_BITS)        /* position in the process's rand() spawn sequence */     \
  X(missile_mask, uint32_t, 1, 0, misc, 8, SF_FK_BITS)        /* bit i = mMissiles[i].mAlive */                         \
  X(shell_mask, uint32_t, 1, 0, misc, 12, SF_FK_BITS)         /* bit i = mShells[i].mAlive */                           \
  X(ep_return, int32_t, 1, 0, timers_a, 12, SF_FK_EPRET)         /* running sum of wrapper rewards (rl/train.py:84) */     \
  X(ep_kills, int32_t, 1, 0, misc, 12, SF_FK_BITS)            /* running sum of info (rl/train.py:81) */                \
  X(ship_angle, int16_t, 1, 0, small, 0, SF_FK_PLAIN)         /* mShip.mAngle: always an integer in [0,360) */          \
  X(fort_angle, int16_t, 1, 0, small, 2, SF_FK_PLAIN)         /* mFortress.mAngle: multiple of the sector size */       \
  X(fort_last_angle, int16_t, 1, 0, small, 4, SF_FK_PLAIN)    /* mFortress.mLastAngle */                                \
  X(missile_angle, int16_t, SF_NSLOT, 0, missile_meta, 0, SF_FK_MPOOL) /* mMissiles[i].mAngle (vel = 20*(cos,sin)) */   \
  X(flags, uint8_t, 1, 0, small, 6, SF_FK_PLAIN)              /* SF_FL_* bits */                                        \
  X(last_reward, int8_t, 1, 0, small, 7, SF_FK_PLAIN)         /* (int)mReward of the last tick = what step_one_tick returned (SRC/game.cpp:484) */

enum SfFieldId {
#define X(name, ctype, count, isf, group, off, kind) SF_F_##name,
  SF_FIELDS(X)
#undef X
      SF_F_COUNT
};

// Packed words.  The 13 statistics of SRC/game.hh:29-43 and the two episode sums are per-episode counts (zeroed with a new
// game); an episode has 5 295 ticks, a key can be pressed once per two ticks, a ship dies at most once per 31 ticks and a
// fortress once per ~90, so 8 / 12 / 16 bits hold them -- and the words of the lane's chunks have that much room to spare:
// the key timers move by one per tick (+-5 295), vlner by one per hit, time is 34 ms per tick, the spawn cursor indexes a
// table of at most 2^24 entries.  No atomics, no counter rows, no bytes of their own.
//   word                      bits  0..                                   above
//   timers_a.0  SF_W_PVL      prev_vlner 12                               vlner_incs 12 (<< 12), big-hex deaths 8 (<< 24)
//   timers_a.4  SF_W_FIRE     fire timer, int16                           resets 16 (<< 16)
//   timers_a.8  SF_W_THRUST   thrust timer, int16                         missed 16 (<< 16)
//   timers_a.12 SF_W_LEFT     left timer, int16                           ep_return bits 0..15 (<< 16)
//   timers_b.0  SF_W_RIGHT    right timer, int16                          ep_return bits 16..31 (<< 16)
//   score.8     SF_W_VLNER    vlner 12                                    max_vlner 12 (<< 12), small-hex deaths 8 (<< 24)
//   score.12    SF_W_TIME     time 24 (ms)                                shell deaths 8 (<< 24)
//   misc.4      SF_W_CURSOR   spawn cursor 24                             destroyed 8 (<< 24)
//   misc.8                    missile alive mask 20                       the tile's missile pool count 12 (<< 20)
//   misc.12                   shell alive mask 20                         ep_kills 8 (<< 24)
//   small.8..15               shots, thrusts, lefts, rights as four uint16 (SF_KEYCOUNT_BYTE)
// Ship deaths (stats[3]) = big-hex + small-hex + shell deaths: killShip has exactly those three call sites
// (SRC/game.cpp:339,345,413), so the sum is not stored.  Limits that follow: a key timer saturates nowhere but wraps
// after 32 767 ticks without an edge of its key (six episodes' worth; every new game zeroes it), `time` after 2^24 ms
// (4.6 hours of game time without a new game: auto_reset off and a caller that never resets).
// sf_get_field / sf_set_field see plain ints: SF_BITFIELDS says where each one lives.
#define SF_KEYCOUNT_BYTE 8
#define SF_ST_KEY_FIRST 7 /* SF_ST_SHOTS */
#define SF_ST_KEY_COUNT 4
// B(field, shift, bits, is_signed): the bit field of the word at the field's (group, byte offset)
#define SF_BITFIELDS(B)          \
  B(prev_vlner, 0, 12, 0)        \
  B(fire_timer, 0, 16, 1)        \
  B(thrust_timer, 0, 16, 1)      \
  B(left_timer, 0, 16, 1)        \
  B(right_timer, 0, 16, 1)       \
  B(vlner, 0, 12, 0)             \
  B(time, 0, 24, 0)              \
  B(spawn_cursor, 0, 24, 0)      \
  B(missile_mask, 0, 20, 0)      \
  B(shell_mask, 0, 20, 0)        \
  B(ep_kills, 24, 8, 0)
// misc words: the two alive masks use SF_MASK_BITS bits; above them ride the tile's pool count (the same value in
// every lane of the tile, so that any lane's chunk tells how many pool entries are live) and ep_kills
#define SF_MASK_BITS 20
#define SF_MASK_LOW 0xFFFFFu
#define SF_MPOOL_SHIFT 20 /* missile word >> 20 = live entries of the tile's missile pool, <= 64 * SF_NSLOT = 1280 */
#define SF_KILLS_SHIFT 24 /* shell word >> 24 = ep_kills */
// missile pool entry meta: heading (integer degrees, 9 bits) | owner lane (6) | slot (5)
#define SF_MM_ANGLE(m) ((m) & 511u)
#define SF_MM_OWNER(m) (((m) >> 9) & 63u)
#define SF_MM_SLOT(m) (((m) >> 15) & 31u)
#define SF_MM_PACK(angle, owner, slot) ((unsigned)(angle) | ((unsigned)(owner) << 9) | ((unsigned)(slot) << 15))

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
constexpr int kTileLanes = 64;  // one wavefront

struct GroupMeta {
  int chunk, slots;
};
constexpr GroupMeta kGroups[SF_G_COUNT] = {
#define G(name, chunk, slots) {chunk, slots},
    SF_GROUPS(G)
#undef G
};
// byte offset, inside a tile, of (group g, slot 0, lane 0)
// (always_inline: with a run-time slot the device compiler otherwise may emit a CALL to these -- a real function with a loop
//  over a table in memory, and a kernel with calls in it: sf_render_kernel once ran 40 % slower for exactly that)
#if defined(__GNUC__) || defined(__clang__)
#define SF_ALWAYS_INLINE __attribute__((always_inline))
#else
#define SF_ALWAYS_INLINE
#endif
SF_ALWAYS_INLINE constexpr long group_offset(int g) {
  long o = 0;
  for (int i = 0; i < g; i++) o += (long)kGroups[i].chunk * kGroups[i].slots * kTileLanes;
  return o;
}
constexpr long kTileBytes = group_offset(SF_G_COUNT);          // 73 728 B
constexpr long kBytesPerLane = kTileBytes / kTileLanes;        // 1152 B
// byte offset, inside a tile, of lane 0's chunk of (group g, slot s)
SF_ALWAYS_INLINE constexpr long chunk_offset(int g, int s = 0) { return group_offset(g) + (long)s * kGroups[g].chunk * kTileLanes; }

struct FieldMeta {
  const char* name;
  int elem_size, count, is_float, group, byte_in_chunk, kind;
};
struct BitField {
  int field, shift, bits, is_signed;
};
constexpr BitField kBitFields[] = {
#define B(name, shift, bits, sgn) {SF_F_##name, shift, bits, sgn},
    SF_BITFIELDS(B)
#undef B
};
constexpr BitField bit_field(int f) {
  for (const BitField& b : kBitFields)
    if (b.field == f) return b;
  return BitField{-1, 0, 32, 0};
}
constexpr FieldMeta kFields[SF_F_COUNT] = {
#define X(name, ctype, count, isf, group, off, kind) {#name, (int)sizeof(ctype), count, isf, SF_G_##group, off, kind},
    SF_FIELDS(X)
#undef X
};
}  // namespace sfl

// Host-built constant block (doubles):
//   [0, 720)    cos/sin(deg2rad(k)) interleaved, k = 0..359 -- indexed per lane, so every
//               workgroup stages it into LDS (SF_LDS_DOUBLES)
//   [720, 744)  big hexagon: per edge (nx, ny, px, py)   SRC/hexagon.cpp:36-48
//   [744, 768)  small hexagon, same -- checked against the constants compiled into the kernels
#define SF_LDS_TRIG 0
#define SF_LDS_BIGHEX 720
#define SF_LDS_SMALLHEX 744
#define SF_CONST_ATAB 768 /* atan(k / 16), k = 0..16 (+ one pad): the table of the kernel's atan2 (sf_atan2_core) */
#define SF_ATAB_DOUBLES 18
#define SF_CONST_DOUBLES (SF_CONST_ATAB + SF_ATAB_DOUBLES)
#define SF_LDS_DOUBLES 720 /* what a workgroup stages into LDS: the cos/sin table (indexed per lane) */

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
  unsigned char* state;      // base of the tiled state block
  long lanes;                // padded lane count (multiple of 256)
  int n_envs;                // real lanes
  const double* consts;      // SF_CONST_DOUBLES doubles in HBM
  const int16_t* spawn;      // [spawn_len][4] (x, y, angle, 0)
  unsigned spawn_mask;       // spawn_len - 1
  unsigned long long action_keys; // 4 bits per action index, up to 16 actions
  int n_actions;
  double start_vx, start_vy; // cos/sin(deg2rad(-60)) as the host's libm gives them (SRC/configs.cpp:43-44)
  // observation
  int obs_type, obs_f64, real_shell_count, obs_dim, auto_reset;
  // episode accumulators / error counter (device)
  unsigned long long* acc;   // SF_ACC_WORDS words: the episode statistics, then the two sticky error counters
  unsigned long long* dbg;   // SF_STAMPS diagnostic builds only: [wave][16] clock stamps; else null
  unsigned* events;          // optional per-tick event bitmask output (SF_EV_*), [n_steps][n_envs]; else null
  // optional trainer bookkeeping of rl/train.py:82-88 + rollouts.insert (sf_step_record): reward as float and
  // mask = 1 - done for this tick ([n_steps][n_envs]); episode / final reward accumulators and the action as
  // int64 per env; t_reward == null switches the whole block off
  float* t_reward;
  float* t_mask;
  float* t_episode;
  float* t_final;
  long long* t_actions;
  // optional VecNormalize reduction (sf_step_normalize): per-wave partial sums of the observations just written
  // and of the discounted returns, column-major [2 * (obs_dim + 1)][lanes / 64]; null = off
  double* n_partials;
  double* n_ret;    // per-env discounted return, ret = ret * n_gamma + reward
  double n_gamma;
  // image batches only: per tile, the envs whose ship died in the last tick (sf_render_kernel's launch order); else null
  unsigned long long* hint;
  // optional: the action every env played this tick as uint8 [n_steps][n_envs] (the sampled ones of sf_step_sampled); else null
  unsigned char* act_out;
  // image batches only: the envs' draw records (sf_drawrec.h), SF_DR_BYTES per lane, written by the step kernel for the frame
  // kernel; else null.  draw_pics: the batch's render pictures exist (the records' decisions depend on it)
  unsigned char* draw;
  int draw_pics;
  // SF_FLAG_REF_RESET_OBS: the observation of a NEW game (sf_reset, the auto-reset of a finished lane) carries aim = vdir =
  // ndist = 0, what the reference's wrapper returns on fresh memory (Game::Game leaves mExtra unwritten, SRC/game.cpp:78)
  int ref_reset_obs;
};

// words of SfKernelArgs::acc behind the SF_EPISODE_STATS_LEN (8) episode statistics
#define SF_ACC_BAD_ACTION 8 /* actions outside [0, n_actions) that ran as NOOP (sf_check_actions) */
#define SF_ACC_OVERFLOW 9   /* times a packed per-episode field of some env left its bits: counted on the tick it wraps (sf_check_state; sticky until sf_reset) */
#define SF_ACC_HANDOVER 10 /* split launches (sf_step_kernel<..., 1000 + BLK>): polls of a hand-over word that gave up (sf_check_state; never seen) */
#define SF_ACC_WORDS 11

// act_type of the step kernel when the lanes draw their own actions (sf_step_sampled): `actions` then points at one
// record per tile, (tick, key0, key1, the tile's first lane in the whole job) -- sf_kernels.hip: sf_philox4x32_10
#define SF_ACT_SAMPLED 0
