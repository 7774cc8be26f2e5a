// sf_kernels.hip -- the env.step() hot path as HIP kernels for gfx950 (MI355X, CDNA4).
//
// One lane per environment.  A launch of sf_step_kernel advances every env of the
// batch by one 34 ms tick and does, fused, what the reference spreads over three
// layers and N processes:
//   SSF_Env.step          ENV:208-253   action -> key events, reward shaping, obs
//   Game::stepOneTick     SRC/game.cpp:473-485 and everything it calls
//   vec-env worker loop   rl/train.py:80 (auto-reset on done)
// (SRC = python/spacefortress/src, ENV = python/spacefortress.gym/.../ssf_env.py of
// the reference.)  The step order, the float32 score adds and the double position
// arithmetic follow the reference operation by operation; the file is compiled
// with -ffp-contract=off so that a*b+c rounds twice, as it does in the reference
// build (baseline x86-64, no FMA).
//
// Hardware mapping:
//  * state is struct-of-arrays per 64-env wave tile in HBM (sf_layout.h): a wave's 64
//    lanes read 64 consecutive elements of each field -- every access is a full
//    coalesced row -- and the whole state of the wave hangs off ONE scalar base with
//    compile-time field offsets (the batch-wide SoA first tried spent a quarter of its
//    instructions on per-field 64-bit address arithmetic and SGPR spills);
//  * at 65 536 envs a launch is 1024 waves = ONE wave per SIMD of the chip, so the
//    kernel is a latency chain, not a throughput problem.  It is organised so that a
//    wave makes two memory round trips, not twenty: (1) every unconditional load is
//    issued up front; (2) as soon as the two alive-bitmasks arrive, the live
//    projectile slots are prefetched -- a wave ballot skips slot groups no lane uses --
//    and their latency hides under the key / ship / fortress arithmetic;
//  * projectile ballistics run as straight-line code over the prefetched slots (the
//    slots are independent until a hit), so their f64 chains interleave; only the
//    rare hit / miss events walk the fortress state machine, in slot order;
//  * the 360-entry cos/sin table (indexed per lane) is staged into LDS once per workgroup (6 KB),
//    BEFORE the predicated loads are issued (any wait behind them is a vmcnt(0)); the hexagon
//    edges and the scalar presets are immediates, what round trip 1 needs of the kernel arguments
//    is preloaded into SGPRs by the command processor;
//  * a lone wave waits out every hand-off between units, so truth values stay in the vector ALU
//    (integer masks, min / max folds, selects instead of small exec-mask branches), constants
//    used per slot live in VGPRs, divisions by constants are three operations;
//  * the four waves of a CU share one address unit: projectile slots, lane chunks and counters go
//    through buffer instructions on a per-wave descriptor (constant part of the address in the
//    scalar offset, idle lanes out of range instead of branched around), what is written once
//    per projectile is one store with the slot in the lane offset;
//  * the per-episode statistics ride in spare bits of words the lane loads and stores anyway (sf_layout.h: SF_W_*): no
//    counter rows, no atomics on the tick; only an episode's END adds its totals to the batch's accumulators with atomics
//    (once per env and 5 295 ticks);
//  * image batches: the tick also leaves the env's draw record for the frame kernel (sf_drawrec.h);
//  * observations are transposed through LDS and leave as 16-byte coalesced stores; every 16-byte
//    store is write-through (sc1), so the end-of-kernel write-back has little left to flush;
//  * no dense contraction anywhere on this path: no MFMA.  The bound is HBM traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#include "sf_internal.h"
#include "sf_layout.h"
#include "sf_deg_dd.h"
#include "sf_drawrec.h"

// Four waves per workgroup share one LDS copy of the cos/sin table (one barrier, early, while the
// waves are still in step; a copy per wave was tried: 1024 waves pulling the same 45 cache lines
// out of L2 at once made the load phase 3x longer).  Everything after that is wave-private.
#ifndef SF_BLOCK
#define SF_BLOCK 256
#endif
// LDS map (doubles): [0, SF_LDS_DOUBLES) the cos/sin table; then one (hit, out) pair of 32-bit event words per lane,
// through which the missile pool's entries tell their owner lanes what happened to them; then the observation staging
#define SF_LDS_EV SF_LDS_DOUBLES
/* then (sf_step_kernel: kLdsAtab, kLdsStage) behind the BLK event words: atan(k / 16), k = 0..16 (sf_atan2_core), and the staging rows */
#define SF_MAX_MISSILES_D 20.0 /* sf.MAX_MISSILES / sf.MAX_SHELLS as divisors (ENV:124-125) */
#ifndef SF_MROWS
#ifndef SF_SPLIT
#define SF_SPLIT 1 /* 1: batches up to 65 536 envs (at most one games' wave per SIMD) step by split launches (sf_step_kernel, BLKP = 1000 + BLK): A/B 4 096 envs 5.48 -> 5.29 us, 32 768: 6.11 -> 5.89, 65 536: 6.51 -> 6.39; 0: never; 2: every batch the instantiation can serve (tests) */
#endif
#define SF_MROWS 3 /* rows of the tile's missile pool (64 entries each) loaded up front with the lane's chunks; more live
                      missiles than that (> 192 in 64 envs; random play averages 104) take the dependent-load loop */
#endif
#define SF_SPF 6 /* shell slots prefetched (groups of SF_SGSZ) */
#ifndef SF_SGSZ
#define SF_SGSZ 1 /* A/B at 65 536 envs with the missile pool: pairs 6.82 us per launch, singles 6.75 */
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// Diagnostic build only (-DSF_STAMPS, tools/stamps.py): shader-clock stamps at phase boundaries,
// with forced waits so each phase owns its memory latency.  The product build has none of it.
#if defined(SF_STAMPS) && !defined(SF_STAMPS_LITE)
#define SF_STAMP(k, drain)                                                 \
  do {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    stamp_[k] = __builtin_amdgcn_s_memtime();                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
  } while (0)
#else
#define SF_STAMP(k, drain)
#endif

// Field access inside the wave's tile (sf_layout.h): `tb` is the tile base -- wave-uniform, one
// SGPR pair for the whole state -- the group/slot offset is a compile-time constant and the lane
// contributes a 32-bit byte offset (one VGPR per chunk size: 16, 8, 4 or 2 bytes).
#define SF_CHUNK(group, s)                                   \
  (tb + sfl::chunk_offset(SF_G_##group, 0) +                 \
   (size_t)(s) * (size_t)(sfl::kGroups[SF_G_##group].chunk * sfl::kTileLanes))
#define SF_LD(T, base, off) (*reinterpret_cast<const T*>((base) + (off)))
// SF_STORE_MODE (A/B switch, tools/ab.py on one device, 65 536 envs): 0 plain stores 11.20 us per
// launch, 1 non-temporal 11.01, 2 write-through (`sc1`) for the 16-byte chunks 10.83 -- the bytes
// leave L2 while the kernel still runs, so the end-of-kernel write-back has less to flush.  (With
// the earlier 1-8-byte rows `sc1` LOST 1 %: narrow write-through stores are one fabric write each.)
// Nothing stored this way is read again inside the launch.
#ifndef SF_OBS_NT
#define SF_OBS_NT 0
#endif
// SF_ABL_TRIG (timing-only ablation, results are WRONG when set): 1 = no workgroup barrier after the
// table staging, 2 = no table at all (hardware v_cos_f32 / v_sin_f32 instead of the exact entries).
#ifndef SF_ABL_TRIG
#define SF_ABL_TRIG 0
#endif
#ifndef SF_AXIS_VEL
#define SF_AXIS_VEL 0 /* 1: the near-axis form of sf_atan2 for the velocity bearing as well.  It only feeds the `vdir`
                         observation, a velocity component within 2^-27 of zero takes an exact cancellation of thrusts that
                         the 0.3 * cos / sin(6k degrees) steps do not produce, and the test costs 0.1 us per launch */
#endif
#ifndef SF_OBS_SC1
#define SF_OBS_SC1 1 /* the observation rows leave write-through (`sc1`) like the state chunks: 5 MB less for the
                        end-of-kernel write-back, 8.19 -> 7.94 us per launch; the 4- and 1-byte outputs gain nothing */
#endif
#ifndef SF_ABL_STATS
#define SF_ABL_STATS 0
#endif
#ifndef SF_ABL_SPAWN
#define SF_ABL_SPAWN 0
#endif
// SF_ABL_PROJ (timing-only ablation, results are WRONG when set): 1 = no missile prefetch / ballistics, 2 = no shells
// either, 3 = only the shells removed.  SF_ABL_OBS: 1 = no observation epilogue.  SF_ABL_ATAN: 1 = no velocity bearing,
// 2 = neither bearing (cheap stand-ins).
#ifndef SF_ABL_PROJ
#define SF_ABL_PROJ 0
#endif
#ifndef SF_ABL_OBS
#define SF_ABL_OBS 0
#endif
#ifndef SF_ABL_ATAN
#define SF_ABL_ATAN 0
#endif
#if SF_ABL_TRIG == 2
#define SF_COS(ang) ((double)__builtin_amdgcn_cosf((float)(ang) * (1.0f / 360.0f)))
#define SF_SIN(ang) ((double)__builtin_amdgcn_sinf((float)(ang) * (1.0f / 360.0f)))
#else
#define SF_COS(ang) trig[2 * (ang)]
#define SF_SIN(ang) trig[2 * (ang) + 1]
#endif
#ifndef SF_STORE_MODE
#define SF_STORE_MODE 2
#endif
// the cache bits of the write-through stores: inline-asm text and the builtins' aux value (1 = sc0, 2 = nt, 16 = sc1)
#ifndef SF_SC_AUX
#define SF_SC_AUX 16
#define SF_SC_ASM "sc1"
#endif
template <typename T>
__device__ __forceinline__ void sf_store(T* p, T v) {
#if SF_STORE_MODE == 1
  __builtin_nontemporal_store(v, p);
#elif SF_STORE_MODE == 2
  if constexpr (sizeof(T) == 16) {
    // the s_nop belongs to the store: a store of more than 64 bits reads its data registers over several cycles and
    // the next VALU write of one of them needs a wait state in between, which the compiler cannot insert for an
    // instruction it does not see (found as misc.prev_vlner of lanes 12-15 of every 16 holding the NEXT store's word)
    asm volatile("global_store_dwordx4 %0, %1, off " SF_SC_ASM "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  } else {
    *p = v;
  }
#else
  *p = v;
#endif
}
#define SF_ST(T, base, off, v) sf_store<T>(reinterpret_cast<T*>((base) + (off)), (T)(v))

// The projectile slots, the lane's chunks and the counters go through `buffer_*` instructions on a per-wave descriptor of the
// tile.  The slot's chunk offset (a compile-time constant too big for the 12-bit immediate) rides in
// the scalar offset instead of costing two 64-bit VALU adds per access, and a lane that has nothing in
// the slot gets an out-of-range offset: the hardware range check returns 0 for its load and drops its
// store, so there is no exec-mask branch around each access.
#define SF_GOFF(group, s) \
  ((unsigned)sfl::chunk_offset(SF_G_##group, 0) + (unsigned)(s) * (unsigned)(sfl::kGroups[SF_G_##group].chunk * sfl::kTileLanes))
#define SF_OOB 0x80000000u /* beyond any tile: the lane's access does not happen */

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

// A 128-bit buffer store with the wait state its data registers need ATTACHED.  A store of more than 64 bits reads its data
// VGPRs over several cycles, and a VALU write of one of them in the next issue slot changes what lanes 12-15 of every 16
// store.  The compiler inserts the s_nop for global / flat stores and for buffer stores with an immediate soffset, but takes a
// buffer store whose soffset is an SGPR to be safe (LLVM GCNHazardRecognizer::createsVALUHazard).  On MI355X that holds
// while the wave is alone on its SIMD -- every batch up to 65 536 envs, every test of rounds 1-3 -- and does not with two or
// more: a batch of 262 144 envs played different games than the same envs in four batches, in exactly those lanes (round 4;
// tests/test_gpu_parity.py::test_batches_beyond_one_wave_per_simd).  So the store goes out as inline assembly with its s_nop
// (the compiler cannot place anything in between), like sf_store's global one; tools/store_hazard_scan.py checks a build's
// assembly for wide buffer stores the compiler emitted bare.  AUX as the builtin's: 0 plain, 16 write-through (sc1).
template <int AUX>
__device__ __forceinline__ void sf_buf_st128(u4_t v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  static_assert(AUX == 0 || AUX == 16 || AUX == 1 || AUX == 17, "cache bits of the store");
  if constexpr (AUX == 16)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else if constexpr (AUX == 1)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else if constexpr (AUX == 17)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
  else
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

namespace {

struct Lane {
  double sx, sy, vx, vy;
  int angle;
  unsigned fl;
  int death_t, fire_t, thrust_t, left_t, right_t;
  int fort_t, fort_death_t, fort_vuln_t;
  int fort_angle, fort_last;
  float points, raw;
  int vlner, time;
  int prev_vlner;
  unsigned cursor, mmask, smask;
  unsigned kc0, kc1;  // key-press counters: shots | thrusts << 16, lefts | rights << 16 (sf_layout.h: SF_KEYCOUNT_BYTE)
  // the per-episode counters that ride above the timers, vlner, time and the cursor (sf_layout.h: SF_W_*)
  int ep_return;
  unsigned c_resets, c_missed, c_incs, c_maxv, c_big, c_small, c_shell, c_destroyed;
  unsigned mpool;     // live entries of the tile's missile pool (wave-uniform; rides above the missile mask)
  unsigned ep_kills;  // sum of info over the episode (rides above the shell mask)
};

// What one tick adds to the statistics (SRC/game.hh:29-43); flushed with atomics.
// (Named scalars, not an array: the compiler merges `if (c) d[5]++; else d[4]++;` into a
// dynamically indexed update, which would push an array into scratch memory.)
struct StatDelta {
  int big_hex_deaths = 0, small_hex_deaths = 0, shell_deaths = 0, ship_deaths = 0, resets = 0, destroyed = 0,
      missed = 0, shots = 0, thrusts = 0, lefts = 0, rights = 0, vlner_incs = 0;
  int max_vlner = 0;  // candidate for counter 12 (a running maximum)
};

struct Off {  // 32-bit byte offsets of this lane into rows of 16-, 8-, 4-, 2- and 1-byte chunks
  unsigned o16, o8, o4, o2, o1;
};

// a / C for a compile-time constant C, bit-identical to the IEEE division it replaces, in three dependent
// operations instead of the eleven of the general v_div_* sequence (v_rcp_f64 included): with rc = RN(1/C),
// q = RN(a * rc) is within an ulp of a / C, rem = a - C * q is exact in an FMA, and RN(q + rem * rc) is the
// correctly rounded quotient (Markstein's theorem).  Checked exhaustively enough on the host for every C used
// here -- pi, 10, 20, 80, 90, 92, 180, 360, 5294: 4e8 operands each, none differ (tests/test_div_const.py keeps a
// smaller run of the same check).  Only the sign of a zero quotient can differ (-0.0 / C gives +0.0 here); nothing
// downstream looks at it.  The operands are angles, pixels and tick counts: no overflow, underflow or NaN.
__device__ __forceinline__ double sf_div_const(double a, double c, double rc) {
  const double q = a * rc;
  const double rem = __builtin_fma(-c, q, a);
  return __builtin_fma(rem, rc, q);
}
#define SF_DIV(a, C) sf_div_const((a), (double)(C), 1.0 / (double)(C))
__device__ __forceinline__ double rad2deg(double a) { return SF_DIV(a, M_PI) * 180; }  // SRC/vector.cpp:38-40
__device__ __forceinline__ double deg2rad(double a) { return SF_DIV(a * M_PI, 180); }  // SRC/vector.cpp:34-36

// atan2 as the reference's libm rounds it where it matters.  The device libm (ocml) is faithful, glibc is correctly
// rounded, and for nearly every argument the last-bit difference is invisible: the results only feed ceil-to-10
// degrees (fortress sector), ceil-to-1 degree (autoturn heading) and observations.  Ships move on near-lattices,
// though (integer spawns, velocities that are sums of 0.3 * cos(6k degrees)), and do cross x = 355 or y = 315 within
// 1e-13: the bearing is then a whisker off +-90 or +-180 degrees -- multiples of 10 -- and which side of the
// boundary the ROUNDED value falls on is decided by that last bit (found by a 3e8-step soak: sector 280 against the
// reference's 270).  Next to the y axis and to the negative x axis the result is therefore formed as
// +-pi/2 - x/y and +-pi + y/x with pi in two doubles: one rounding, the correctly rounded value, bit for bit what
// glibc returns there (4e7 such arguments checked on the host, tests/native/atan2_axis.c).  A wave-wide test skips
// the block on all but a handful of ticks.
//
// The same last bit decides whenever the bearing is within rounding noise of ANY integer degree, and in autoturn games
// that is a regime, not an accident: a ship that thrusts at the fortress flies along an exact-degree ray.  RAZOR = 1:
// within 1e-9 degrees of k degrees the result is formed as phi_k + N / D, N = |y| cos k - x sin k in double-double
// (products exact by FMA; phi, cos, sin of k = 0..180 as (hi, lo) pairs, sf_deg_dd.h), D = x cos k + |y| sin k: the
// correctly rounded value.  glibc's own atan2 is not correctly rounded in 0.08 % of such arguments (0.503-ulp errors,
// tools/atan2_razor), so agreement there is 99.9 %, not 100 % -- against a coin toss per tick for the plain device libm.
// atan2 for the step kernel's hot path, half the instructions of the device libm's: no special cases (the arguments are
// finite coordinate / velocity differences), the quotient q = min / max in [0, 1] by a reciprocal and Newton steps, then
// one table step atan(q) = atan(k / 16) + atan(t), t = (q - k/16) / (1 + q k/16), |t| <= 1/32, where five terms of the
// series leave 3e-18.  `atab` = atan(k / 16), k = 0..16, in LDS (host libm, sf_host_fill_consts).  Within 1e-15 rad of
// the host libm's atan2 (a few ulps; tests/native/atan2_core.c restates it on the host); everything that needs MORE than
// that -- the axes, the integer degrees -- is decided by sf_atan2's exact forms below, which do not look at this value's
// last bits.  (0, 0) gives a NaN: the only caller that can pass it, the velocity bearing, discards the value for a ship
// at rest.
__device__ __forceinline__ double sf_recip(double d) {  // 1 / d to an ulp or so, d in the normal range
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}
__device__ __forceinline__ double sf_atan2_core(double y, double x, const double* atab) {
  const double ax = fabs(x), ay = fabs(y);
  const double u = __builtin_fmax(ax, ay), v = __builtin_fmin(ax, ay);
  const double ru = sf_recip(u);
  double q = v * ru;
  q = __builtin_fma(__builtin_fma(-u, q, v), ru, q);
  const double k = rint(q * 16.0), c = k * 0.0625;
  const double den = __builtin_fma(q, c, 1.0), num = q - c;
  const double rd = sf_recip(den);
  double t = num * rd;
  t = __builtin_fma(__builtin_fma(-den, t, num), rd, t);
  const double s = t * t;
  double p = __builtin_fma(s, 1.0 / 9.0, -1.0 / 7.0);
  p = __builtin_fma(s, p, 0.2);
  p = __builtin_fma(s, p, -1.0 / 3.0);
  double a = atab[(int)k] + __builtin_fma(t, p * s, t);
  a = ay > ax ? 1.5707963267948966 - a : a;
  a = x < 0 ? 3.141592653589793 - a : a;
  return copysign(a, y);
}
#ifndef SF_FAST_ATAN
#define SF_FAST_ATAN 1 /* 0: the device libm's atan2 everywhere (A/B) */
#endif

template <bool RAZOR>
__device__ __forceinline__ double sf_atan2(double y, double x, const double* atab = nullptr) {
  double r = (SF_FAST_ATAN && atab) ? sf_atan2_core(y, x, atab) : atan2(y, x);
  const double ax = fabs(x), ay = fabs(y);
  const bool ny = ax * 0x1p27 < ay;              // next to the y axis (x == 0 included)
  const bool nx = (x < 0) & (ay * 0x1p27 < ax);  // next to the negative x axis (y == 0 included)
  if (__ballot(ny | nx) != 0ull) {
    if (ny | nx) {
      const double t = ny ? x / y : y / x;
      const double hi = ny ? 1.5707963267948966 : 3.141592653589793;          // pi/2, pi
      const double lo = ny ? 6.123233995736766e-17 : 1.2246467991473532e-16;  // their low parts
      r = copysign(hi, y) + (ny ? copysign(lo, y) - t : copysign(lo, y) + t);
    }
  }
  if (RAZOR) {
    const double deg = SF_DIV(fabs(r), M_PI) * 180, kd = rint(deg);
    const bool rz = !(ny | nx) & (fabs(deg - kd) < 1e-9) & (y != 0.0);
    if (__ballot(rz) != 0ull) {
      if (rz) {
        const double* e = kDegDD[(int)kd];  // (phi_hi, phi_lo, cos_hi, cos_lo, sin_hi, sin_lo) of kd degrees
        const double ph = e[0], pl = e[1], ch = e[2], cl = e[3], sh = e[4], sl = e[5];
        const double p1 = ay * ch, e1 = __builtin_fma(ay, ch, -p1);
        const double p2 = x * sh, e2 = __builtin_fma(x, sh, -p2);
        const double d = p1 - p2;  // nearly cancels
        const double bb = d - p1, err = (p1 - (d - bb)) + (-p2 - bb);  // two-sum error term of p1 + (-p2)
        const double lo = err + (e1 - e2) + (ay * cl - x * sl);
        const double N = d + lo, D = x * ch + ay * sh;
        r = copysign(ph + (pl + N / D), y);
      }
    }
  }
  return r;
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 constants): counter
// (c0, c1, 0, 0), key (k0, k1); the first output word.  The reference's rollout gets its actions from the policy
// (rl/train.py:76-80); the random-action benchmark draws them here: action = floor(x * n_actions / 2^32).
__device__ __forceinline__ unsigned sf_philox4x32_10(unsigned c0, unsigned c1, unsigned k0, unsigned k1) {
  unsigned c2 = 0u, c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
    const unsigned lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

// Game::reward (SRC/game.cpp:97-102): three float32 adds in this order, points clamped at 0.
__device__ __forceinline__ void score(float amount, float& rew, Lane& L) {
  rew += amount;
  L.raw += amount;
  L.points += amount;
  if (L.points < 0) L.points = 0;
}

// Hexagon::isInside (SRC/hexagon.cpp:36-48).  The edges (nx, ny, px, py) are compile-time
// constants (sf_layout.h: both radii are the same in every preset), so they are immediates.
// For the two horizontal edges of each hexagon nx is -0.0, so  nx*dx + ny*dy < 0  is exactly
// ny*dy < 0 (adding a zero changes nothing, a zero product is not < 0), i.e. a plain comparison
// of y against the edge: y < py for ny > 0, y > py for ny < 0 (the sign of a difference of two
// doubles is exact, and scaling by |ny| >= 1 cannot flush it to zero).  Eight multiplies, eight
// subtractions and four additions less per ship and tick, same truth value for every finite y.
// The six truth values are folded arithmetically: "no edge value is < 0" is "the smallest edge value is not < 0"
// (for the horizontal edges the value is the difference y - py or py - y; -0.0 is not < 0 either way; no NaN
// here), one comparison per hexagon instead of six whose results meet in scalar registers.
#define SF_EDGE_TEST(nx, ny, px, py)                                                                  \
  m = __builtin_fmin(m, (nx) == 0.0 ? ((ny) > 0 ? y - (py) : (py) - y) : (nx) * (x - (px)) + (ny) * (y - (py)));
__device__ __forceinline__ bool inside_big_hex(double x, double y) {
  double m = 1.0;
  SF_BIG_HEX_EDGES(SF_EDGE_TEST)
  return !(m < 0);
}
__device__ __forceinline__ bool inside_small_hex(double x, double y) {
  double m = 1.0;
  SF_SMALL_HEX_EDGES(SF_EDGE_TEST)
  return !(m < 0);
}
#undef SF_EDGE_TEST
#define SF_EDGE_TEST(nx, ny, px, py)                                     \
  if ((nx) == 0.0)                                                       \
    in = in & !((ny) > 0 ? (y < (py)) : (y > (py)));                     \
  else                                                                   \
    in = in & !((nx) * (x - (px)) + (ny) * (y - (py)) < 0);
#undef SF_EDGE_TEST

// Game::isOutsideGameArea (SRC/game.cpp:129-131)
__device__ __forceinline__ bool outside_area(const SfKernelArgs& a, double x, double y) {
  return (x < 0) | (x > sfc::width_d) | (y > sfc::height_d) | (y < 0);
}

// Game::resetShip (SRC/game.cpp:133-149).  The accepted (x, y, angle) of the rejection loop over
// libc rand() is a fixed sequence per seed: the host precomputed it (sf_spawn_table) and each
// lane walks it with its own cursor.
// `e` = the lane's next table entry, packed (x, y, angle, 0) as four int16
__device__ __forceinline__ void spawn_ship_from(const SfKernelArgs& a, Lane& L, unsigned long long e) {
  L.cursor += 1;
  L.sx = (double)(int16_t)(e & 0xFFFFu);
  L.sy = (double)(int16_t)((e >> 16) & 0xFFFFu);
  L.angle = (int16_t)((e >> 32) & 0xFFFFu);
  L.vx = a.start_vx;
  L.vy = a.start_vy;
  L.fl |= SF_FL_SHIP_ALIVE;
}
__device__ __forceinline__ void spawn_ship(const SfKernelArgs& a, Lane& L) {
  spawn_ship_from(a, L, *reinterpret_cast<const unsigned long long*>(a.spawn + 4 * (size_t)(L.cursor & a.spawn_mask)));
}

// Game::Game (SRC/game.cpp:18-82); statistics and episode sums are zeroed by the caller
__device__ __forceinline__ void new_game(const SfKernelArgs& a, Lane& L) {
  L.fl = 0;
  spawn_ship(a, L);
  L.fl |= SF_FL_FORT_ALIVE;
  L.fort_angle = 180;  // :40
  L.fort_last = 0;     // :41
  L.points = 0;
  L.raw = 0;
  L.vlner = 0;
  L.time = 0;
  L.death_t = L.fire_t = L.thrust_t = L.left_t = L.right_t = 0;
  L.fort_t = L.fort_death_t = 0;
  L.fort_vuln_t = sfc::vuln_time;  // :78 adds to a never-initialised member; defined as 0 + 250
  L.mmask = L.smask = 0;
  L.kc0 = L.kc1 = 0;  // statistics start over with the game (SRC/game.cpp:18-82)
  L.ep_return = 0;
  L.c_resets = L.c_missed = L.c_incs = L.c_maxv = L.c_big = L.c_small = L.c_shell = L.c_destroyed = 0;
  L.ep_kills = 0;
  // (L.mpool belongs to the tile, not to the game: the caller maintains it)
}

__device__ __forceinline__ void kill_ship(Lane& L, StatDelta& S) {  // SRC/game.cpp:274-280
  if (L.fl & SF_FL_SHIP_ALIVE) {
    L.fl &= ~SF_FL_SHIP_ALIVE;
    L.death_t = 0;
    S.ship_deaths += 1;
  }
}

// The fixed part of a lane: eight 16-byte chunks (sf_layout.h), in two sets.  The start of a launch is a chip-wide burst --
// every wave of every CU pulls its state at once and the fabric delivers about 12 bytes per cycle and CU -- so what the
// first phases of the tick need (keys, respawn, ship, fortress: flags and angles, masks, the timers, position and
// velocity) is issued FIRST and waited for alone; the two chunks that are first read at the shells or later (score,
// counts) and the missile pool rows are issued behind the dependent loads of round trip 2 and arrive under the
// key / ship / fortress arithmetic.
struct LaneLate {
  i4_t ta, sc;
};
__device__ __forceinline__ void load_lane_early(const unsigned char* tb, const Off& o, Lane& L) {
  const i4_t mi = SF_LD(i4_t, SF_CHUNK(misc, 0), o.o16);  // first: the projectile prefetch waits on the masks
  const i4_t sm = SF_LD(i4_t, SF_CHUNK(small, 0), o.o16);
  const d2_t p = SF_LD(d2_t, SF_CHUNK(ship_pos, 0), o.o16);
  const d2_t v = SF_LD(d2_t, SF_CHUNK(ship_vel, 0), o.o16);
  const i4_t tc = SF_LD(i4_t, SF_CHUNK(timers_b, 0), o.o16);
  L.right_t = (int)(int16_t)(tc.x & 0xFFFF);
  L.ep_return = (int)((unsigned)tc.x & 0xFFFF0000u);  // bits 16..31; the low half comes with the late set
  L.fort_t = tc.y;
  L.fort_death_t = tc.z;
  L.fort_vuln_t = tc.w;
  L.death_t = mi.x;
  L.cursor = (unsigned)mi.y & 0xFFFFFFu;
  L.c_destroyed = (unsigned)mi.y >> 24;
  L.mmask = (unsigned)mi.z & SF_MASK_LOW;
  L.mpool = (unsigned)mi.z >> SF_MPOOL_SHIFT;
  L.smask = (unsigned)mi.w & SF_MASK_LOW;
  L.ep_kills = (unsigned)mi.w >> SF_KILLS_SHIFT;
  L.sx = p.x;
  L.sy = p.y;
  L.vx = v.x;
  L.vy = v.y;
  L.angle = (int16_t)(sm.x & 0xFFFF);
  L.fort_angle = (int16_t)((unsigned)sm.x >> 16);
  L.fort_last = (int16_t)(sm.y & 0xFFFF);
  L.fl = ((unsigned)sm.y >> 16) & 0xFFu;
  L.kc0 = (unsigned)sm.z;
  L.kc1 = (unsigned)sm.w;
}
__device__ __forceinline__ LaneLate load_lane_late(const unsigned char* tb, const Off& o) {
  LaneLate t;
  t.ta = SF_LD(i4_t, SF_CHUNK(timers_a, 0), o.o16);
  t.sc = SF_LD(i4_t, SF_CHUNK(score, 0), o.o16);
  return t;
}
__device__ __forceinline__ void unpack_lane_late(const LaneLate& t, Lane& L) {
  const unsigned w_pvl = (unsigned)t.ta.x, w_fire = (unsigned)t.ta.y, w_thr = (unsigned)t.ta.z, w_left = (unsigned)t.ta.w;
  const unsigned w_vl = (unsigned)t.sc.z, w_time = (unsigned)t.sc.w;
  L.prev_vlner = (int)(w_pvl & 0xFFFu);
  L.c_incs = (w_pvl >> 12) & 0xFFFu;
  L.c_big = w_pvl >> 24;
  L.fire_t = (int)(int16_t)(w_fire & 0xFFFFu);
  L.c_resets = w_fire >> 16;
  L.thrust_t = (int)(int16_t)(w_thr & 0xFFFFu);
  L.c_missed = w_thr >> 16;
  L.left_t = (int)(int16_t)(w_left & 0xFFFFu);
  L.ep_return = (int)((unsigned)L.ep_return | (w_left >> 16));  // the high half came with timers_b
  L.points = __int_as_float(t.sc.x);
  L.raw = __int_as_float(t.sc.y);
  L.vlner = (int)(w_vl & 0xFFFu);
  L.c_maxv = (w_vl >> 12) & 0xFFFu;
  L.c_small = w_vl >> 24;
  L.time = (int)(w_time & 0xFFFFFFu);
  L.c_shell = w_time >> 24;
}

// The lane's seven chunks back to the tile, through the wave's descriptor: the chunk offsets ride in the scalar
// offset, no 64-bit address per store; write-through (SF_SC_AUX).
__device__ __forceinline__ void store_lane_buf(__amdgpu_buffer_rsrc_t rs, const Off& o, const Lane& L) {
  constexpr int aux = SF_STORE_MODE == 2 ? SF_SC_AUX : 0;
#define SF_BST16(group, v) \
  sf_buf_st128<aux>(__builtin_bit_cast(u4_t, v), rs, o.o16, SF_GOFF(group, 0))
  SF_BST16(ship_pos, (d2_t{L.sx, L.sy}));
  SF_BST16(ship_vel, (d2_t{L.vx, L.vy}));
  // the packed words of sf_layout.h (SF_W_*): a value below, a per-episode counter above
  const unsigned er = (unsigned)L.ep_return;
  SF_BST16(timers_a, (i4_t{(int)(((unsigned)L.prev_vlner & 0xFFFu) | ((L.c_incs & 0xFFFu) << 12) | (L.c_big << 24)),
                           (int)(((unsigned)L.fire_t & 0xFFFFu) | (L.c_resets << 16)),
                           (int)(((unsigned)L.thrust_t & 0xFFFFu) | (L.c_missed << 16)),
                           (int)(((unsigned)L.left_t & 0xFFFFu) | (er << 16))}));
  SF_BST16(timers_b, (i4_t{(int)(((unsigned)L.right_t & 0xFFFFu) | (er & 0xFFFF0000u)), L.fort_t, L.fort_death_t, L.fort_vuln_t}));
  SF_BST16(score, (i4_t{__float_as_int(L.points), __float_as_int(L.raw),
                        (int)(((unsigned)L.vlner & 0xFFFu) | ((L.c_maxv & 0xFFFu) << 12) | (L.c_small << 24)),
                        (int)(((unsigned)L.time & 0xFFFFFFu) | (L.c_shell << 24))}));
  SF_BST16(misc, (i4_t{L.death_t, (int)((L.cursor & 0xFFFFFFu) | (L.c_destroyed << 24)), (int)(L.mmask | (L.mpool << SF_MPOOL_SHIFT)),
                       (int)(L.smask | (L.ep_kills << SF_KILLS_SHIFT))}));
#undef SF_BST16
  sf_buf_st128<aux>(
      u4_t{(unsigned)(L.angle & 0xFFFF) | ((unsigned)(L.fort_angle & 0xFFFF) << 16),
           (unsigned)(L.fort_last & 0xFFFF) | ((L.fl & 0xFFFFu) << 16), L.kc0, L.kc1},
      rs, o.o16, SF_GOFF(small, 0));
}

__device__ __forceinline__ void store_lane(unsigned char* tb, const Off& o, const Lane& L) {
  store_lane_buf(__builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)sfl::kTileBytes, 0x00020000), o, L);
}

// Agent-scope (L2-coherent, L1-bypassing) accesses for the few places where one launch may read
// back what it wrote earlier or mixes plain stores with atomics on the same word: the counter
// rows (atomics + the zeroing at a new game) and the dependent-load slow path of the projectile
// slots beyond the prefetched groups (a fused launch reads in tick k+1 what tick k stored).
__device__ __forceinline__ int ld_coherent_i32(const unsigned char* p) {
  return __hip_atomic_load(reinterpret_cast<const int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_coherent_i16(const unsigned char* p) {
  return (int16_t)__hip_atomic_load(reinterpret_cast<const unsigned short*>(p), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ d2_t ld_coherent_d2(const unsigned char* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return d2_t{__longlong_as_double((long long)x), __longlong_as_double((long long)y)};
}
__device__ __forceinline__ void st_coherent_i32(unsigned char* p, int v) {
  __hip_atomic_store(reinterpret_cast<int*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ExtraGameValues of Game::computeExtra (SRC/game.cpp:282-312).  They are a pure function of the
// ship state (frozen while the ship is dead), so they are derived for the observation instead of
// being stored.  a_pos = atan2(sy - fy, sx - fx) is shared with updateFortress, a_vel =
// atan2(vy, vx); both are evaluated side by side so the two dependency chains interleave.
struct Extras {
  double aim, vdir, ndist;
};

__device__ __forceinline__ Extras compute_extras(const SfKernelArgs& a, const Lane& L, double a_pos, double a_vel) {
  Extras e;
  // aim (SRC/game.cpp:299-305)
  double o = rad2deg(a_pos) - (double)L.angle + 180;
  if (o < -180) o = o + 360;
  e.aim = o;
  // vdir (SRC/game.cpp:286-297).  norm()==0 iff vx*vx+vy*vy==0.  The reference's first atan2 is
  // atan2(-(fy-sy), fx-sx) = atan2(dy, -dx) = +-pi - a_pos: derived from a_pos (observation-only
  // value, differs from a second libm call by <= 1 ulp of pi).
  {
    const double dy = L.sy - sfc::fort_y;
    double ov;
    if (dy == 0)  // on the fortress row the two calls sit on different branch cuts: call it
      ov = sf_atan2<false>(-(sfc::fort_y - L.sy), sfc::fort_x - L.sx);
    else
      ov = dy < 0 ? (-M_PI - a_pos) : (M_PI - a_pos);
    double diff = a_vel - ov;
    if (diff > M_PI) diff -= M_PI * 2;
    if (diff < -M_PI) diff += M_PI * 2;
    e.vdir = (L.vx * L.vx + L.vy * L.vy == 0.0) ? 0.0 : rad2deg(diff);
  }
  // fdist, ndist (SRC/game.cpp:310-311): the y term of the reference subtracts the ship from
  // itself, so fdist = sqrt(dx^2 + 0) = |dx|.
  const double fdist = fabs(L.sx - sfc::fort_x);
  e.ndist = -1 + SF_DIV(fdist - sfc::ndist_a, sfc::ndist_b);
  return e;
}

// One observation row (ENV:95-157) written to `o` (an LDS staging row or global memory).
template <typename T>
__device__ __forceinline__ void write_obs(const SfKernelArgs& a, T* o, const Lane& L, const Extras& e) {
  const int n_missiles = __popc(L.mmask);
  const int n_shells = a.real_shell_count ? __popc(L.smask) : n_missiles;  // SRC/pymodule.cpp:131-134
  // ENV:148 reads the vulnerability timer through a getter with undefined behaviour
  // (SRC/pymodule.cpp:44-45); the intended predicate is used.
  const int kill_ready = (L.vlner > 10 && L.fort_vuln_t < sfc::vuln_time) ? 1 : 0;
  const int n_keys_t = a.obs_dim - 15;
  const int timers[4] = {L.fire_t, L.thrust_t, L.left_t, L.right_t};  // SRC/pymodule.cpp:98-105
  const bool ship_alive = L.fl & SF_FL_SHIP_ALIVE, fort_alive = L.fl & SF_FL_FORT_ALIVE;
  if (a.obs_type == 2) {  // monitors, ENV:96-108
    o[0] = (T)(n_missiles > 0 ? 0.5 : -0.5);
    o[1] = (T)(fort_alive ? 0.5 : -0.5);
    o[2] = (T)(L.vlner > 10 ? 0.5 : -0.5);
    o[3] = (T)(kill_ready ? 0.5 : -0.5);
    o[4] = (T)(e.aim < 3 ? 0.5 : -0.5);
    o[5] = (T)(e.aim > 3 ? 0.5 : -0.5);
    o[6] = (T)(e.ndist > .75 ? 0.5 : -0.5);
    o[7] = (T)(e.ndist > .25 ? 0.5 : -0.5);
    o[8] = (T)(e.ndist < -.25 ? 0.5 : -0.5);
    o[9] = (T)(e.ndist < -.75 ? 0.5 : -0.5);
  } else if (a.obs_type == 1) {  // normalized-features, ENV:109-133
    double f[19];
    f[0] = ship_alive ? 1 : 0;
    f[1] = SF_DIV(L.sx, sfc::pb_width);
    f[2] = SF_DIV(L.sy, sfc::pb_height);
    f[3] = SF_DIV(L.vx, 10);
    f[4] = SF_DIV(L.vy, 10);
    f[5] = SF_DIV((double)L.angle, 360);
    f[6] = SF_DIV(e.aim, 180);
    {
      double m = fmod(e.vdir, 360.0);  // Python float %: result takes the divisor's sign
      if (m != 0) {
        if (m < 0) m += 360.0;
      } else {
        m = 0.0;
      }
      f[7] = SF_DIV(m, 360);
    }
    f[8] = e.ndist;
    f[9] = fort_alive ? 1 : 0;
    f[10] = SF_DIV((double)L.fort_angle, 360);
    f[11] = SF_DIV((double)(L.vlner > 10 ? L.vlner : 10), 10);  // ENV:122 max(), as written
    f[12] = kill_ready;
    f[13] = SF_DIV((double)n_missiles, SF_MAX_MISSILES_D);
    f[14] = SF_DIV((double)n_shells, SF_MAX_MISSILES_D);
#pragma unroll
    for (int k = 0; k < 4; k++) f[15 + k] = SF_DIV((double)timers[k], sfc::max_ticks);
#pragma unroll
    for (int k = 0; k < 19; k++) {
      if (k < 15 + n_keys_t) {
        double v = f[k];
        v = v < -1 ? -1 : (v > 1 ? 1 : v);
        o[k] = (T)v;
      }
    }
  } else {  // features, ENV:134-157
    o[0] = (T)(ship_alive ? 1 : 0);
    o[1] = (T)L.sx;
    o[2] = (T)L.sy;
    o[3] = (T)L.vx;
    o[4] = (T)L.vy;
    o[5] = (T)L.angle;
    o[6] = (T)e.aim;
    o[7] = (T)e.vdir;
    o[8] = (T)e.ndist;
    o[9] = (T)(fort_alive ? 1 : 0);
    o[10] = (T)L.fort_angle;
    o[11] = (T)L.vlner;
    o[12] = (T)kill_ready;
    o[13] = (T)n_missiles;
    o[14] = (T)n_shells;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < n_keys_t) o[15 + k] = (T)timers[k];
  }
}

// The default observation -- obs_type 'features', float32, full waves, 16-byte aligned output -- with everything
// the generic path decides at run time fixed at compile time (OBSK = 1 instantiations of the step kernel): DIM
// stores into the wave's staging rows at constant offsets, then the wave's 64 rows (one contiguous span of
// 64 * DIM floats, in LDS and in global memory alike) leave as 16-byte pieces: all LDS reads issued, then all stores.
template <int DIM>
__device__ __forceinline__ void write_features_f32(float* o, const Lane& L, const Extras& e, int real_shell_count) {
  const int n_missiles = __popc(L.mmask);
  const int n_shells = real_shell_count ? __popc(L.smask) : n_missiles;  // SRC/pymodule.cpp:131-134
  const int kill_ready = (L.vlner > 10 && L.fort_vuln_t < sfc::vuln_time) ? 1 : 0;
  const int timers[4] = {L.fire_t, L.thrust_t, L.left_t, L.right_t};
  o[0] = (L.fl & SF_FL_SHIP_ALIVE) ? 1.0f : 0.0f;  // ENV:134-157, as in write_obs
  o[1] = (float)L.sx;
  o[2] = (float)L.sy;
  o[3] = (float)L.vx;
  o[4] = (float)L.vy;
  o[5] = (float)L.angle;
  o[6] = (float)e.aim;
  o[7] = (float)e.vdir;
  o[8] = (float)e.ndist;
  o[9] = (L.fl & SF_FL_FORT_ALIVE) ? 1.0f : 0.0f;
  o[10] = (float)L.fort_angle;
  o[11] = (float)L.vlner;
  o[12] = (float)kill_ready;
  o[13] = (float)n_missiles;
  o[14] = (float)n_shells;
#pragma unroll
  for (int k = 0; k < DIM - 15; k++) o[15 + k] = (float)timers[k];
}
template <int DIM>
__device__ __forceinline__ void flush_features_f32(const float* stage_w, float* dst_w, unsigned lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  typedef float f4_t __attribute__((ext_vector_type(4)));
  constexpr int NV = 64 * DIM / 4, IT = (NV + 63) / 64;  // 304 pieces of 16 bytes (DIM 19), 272 (DIM 17)
  f4_t v[IT];
#pragma unroll
  for (int k = 0; k < IT; k++)
    if (k * 64 + 63 < NV || (int)lane + k * 64 < NV) v[k] = reinterpret_cast<const f4_t*>(stage_w)[lane + k * 64];
#pragma unroll
  for (int k = 0; k < IT; k++)
    if (SF_ABL_OBS != 2 && (k * 64 + 63 < NV || (int)lane + k * 64 < NV)) {  // SF_ABL_OBS 2 (timing-only): no global stores
#if SF_OBS_SC1
      sf_store<f4_t>(reinterpret_cast<f4_t*>(dst_w) + lane + k * 64, v[k]);  // write-through, like the state chunks
#else
      reinterpret_cast<f4_t*>(dst_w)[lane + k * 64] = v[k];
#endif
    }
}

// A wave's 64 observation rows sit in ITS OWN piece of LDS as [lane][obs_dim]; global memory wants
// exactly the same order ([N, obs_dim] row-major), so the wave's rows form one contiguous span:
// copy it with 16-byte lanes-consecutive stores instead of obs_dim strided 4-byte stores per lane.
// Wave-private on purpose: no workgroup barrier, so a fast wave never waits for the divergence
// tail of a slower one (LDS is in-order per wave; the fence only pins the compiler).
template <typename T>
__device__ __forceinline__ void flush_obs_wave(const SfKernelArgs& a, const T* stage_w, T* obs, unsigned wave_env0,
                                               unsigned lane, bool vec_ok) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  long rows = (long)a.n_envs - (long)wave_env0;
  if (rows > 64) rows = 64;
  if (rows <= 0) return;
  const int total = (int)rows * a.obs_dim;
  T* dst = obs + (size_t)wave_env0 * a.obs_dim;
  constexpr int V = 16 / (int)sizeof(T);
  int done_elems = 0;
  if (vec_ok) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int nvec = total / V;
    for (int v = lane; v < nvec; v += 64) {
#if SF_OBS_NT
      __builtin_nontemporal_store(reinterpret_cast<const vec_t*>(stage_w)[v], reinterpret_cast<vec_t*>(dst) + v);
#elif SF_OBS_SC1
      sf_store<vec_t>(reinterpret_cast<vec_t*>(dst) + v, reinterpret_cast<const vec_t*>(stage_w)[v]);
#else
      reinterpret_cast<vec_t*>(dst)[v] = reinterpret_cast<const vec_t*>(stage_w)[v];
#endif
    }
    done_elems = nvec * V;
  }
  for (int t = done_elems + lane; t < total; t += 64) dst[t] = stage_w[t];
}

// VecNormalize's batch sums for this wave's rows, from the observation tile the wave has just flushed (it is
// still in the wave's private piece of LDS): lanes [g*dim, (g+1)*dim) add every groups-th row of feature
// lane % dim and of its square; the tile is then reused to fold the groups.  One row of the column-major
// partial-sum array per wave (sf_normalize.hip adds the rows up); no workgroup barrier.
template <typename T>
__device__ __forceinline__ void norm_partials_wave(const SfKernelArgs& a, T* stage_w, unsigned wave_env0, unsigned lane,
                                                   double my_ret, bool has_obs) {
  const int dim = a.obs_dim, groups = 64 / dim;
  const size_t waves = (size_t)(a.lanes / 64), w = wave_env0 >> 6;
  long rows = (long)a.n_envs - (long)wave_env0;
  rows = rows > 64 ? 64 : (rows < 0 ? 0 : rows);
  double s = 0, q = 0;
  if (has_obs && (int)lane < groups * dim)
    for (int r = (int)lane / dim; r < rows; r += groups) {
      const double v = (double)stage_w[r * dim + (int)lane % dim];
      s += v;
      q += v * v;
    }
  __builtin_amdgcn_wave_barrier();
  double* fold = reinterpret_cast<double*>(stage_w);  // 2 x 64 doubles <= 64 rows x dim x sizeof(T) for dim >= 4
  if (has_obs) {
    fold[lane] = s;
    fold[64 + lane] = q;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if ((int)lane < dim) {
    double ts = 0, tq = 0;
    if (has_obs)
      for (int g = 0; g < groups; g++) {
        ts += fold[g * dim + lane];
        tq += fold[64 + g * dim + lane];
      }
    a.n_partials[(size_t)lane * waves + w] = ts;
    a.n_partials[(size_t)(dim + 1 + lane) * waves + w] = tq;
  }
  double rs = (long)lane < rows ? my_ret : 0.0, rq = rs * rs;
  for (int o = 32; o > 0; o >>= 1) {
    rs += __shfl_xor(rs, o);
    rq += __shfl_xor(rq, o);
  }
  if (lane == 0) {
    a.n_partials[(size_t)dim * waves + w] = rs;
    a.n_partials[(size_t)(2 * dim + 1) * waves + w] = rq;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// VecEnv.reset(): a brand-new Game in every lane (ENV:163-178).  first != 0 additionally
// initialises what SSF_Env.__init__ sets once (prev_vlner, ENV:92) and the spawn cursors.
__global__ __launch_bounds__(SF_BLOCK) void sf_reset_kernel(SfKernelArgs a, int first, unsigned cursor0,
                                                           unsigned cursor_stride, void* obs) {
  const unsigned i = blockIdx.x * SF_BLOCK + threadIdx.x;
  const unsigned lane = threadIdx.x & 63u;
  unsigned char* const tb = a.state + (size_t)__builtin_amdgcn_readfirstlane(i >> 6) * sfl::kTileBytes;
  const Off o = {lane * 16u, lane * 8u, lane * 4u, lane * 2u, lane};
  Lane L;
  if (first) {
    L.prev_vlner = 0;
    L.cursor = cursor0 + cursor_stride * i;
  } else {
    const i4_t mi = SF_LD(i4_t, SF_CHUNK(misc, 0), o.o16);
    L.prev_vlner = SF_LD(int, SF_CHUNK(timers_a, 0), o.o16) & 0xFFF;
    L.cursor = (unsigned)mi.y;
  }
  new_game(a, L);
  L.mpool = 0;  // every env of the tile starts over: the tile's missile pool is empty
  store_lane(tb, o, L);
  if (obs != nullptr && i < (unsigned)a.n_envs && a.obs_type != 3) {
    // the reference's extras are stale heap until the first tick; defined as computeExtra(spawn) -- or, by flag, as the zeros
    // a fresh process's heap holds there (SF_FLAG_REF_RESET_OBS)
    Extras e = compute_extras(a, L, sf_atan2<true>(L.sy - sfc::fort_y, L.sx - sfc::fort_x), sf_atan2<false>(L.vy, L.vx));
    if (a.ref_reset_obs) e = Extras{0.0, 0.0, 0.0};
    if (a.obs_f64)
      write_obs<double>(a, (double*)obs + (size_t)i * a.obs_dim, L, e);
    else
      write_obs<float>(a, (float*)obs + (size_t)i * a.obs_dim, L, e);
  }
}

// ---------------------------------------------------------------------------------------------
// FUSED = false: one tick per launch (sf_step, the VecEnv.step path).
// FUSED = true:  n_steps ticks per launch with the actions of all of them given up front
// (sf_rollout): the wave keeps its state in registers between ticks, so a tick costs neither the
// two memory round trips nor a kernel boundary.  Same body, bit-identical results.
// XTRA = the launch may need what the plain VecEnv.step never does -- actions drawn in the kernel (SF_ACT_SAMPLED), the
// played actions written out (a.act_out), the packed counters' overflow test of batches without auto-reset: three uniform
// tests and their code, 0.07 us of a 6.5 us launch when they sit in the one kernel everybody runs (A/B, tools/ab.py)
// BLK = threads per workgroup (64, 128 or 256: one to four tiles).  256 is what the metric's batch wants (65 536 envs = 256
// workgroups, one per CU, the cos/sin table staged once per four waves); a batch of 16 384 envs is 256 waves all the same, and
// as 64 workgroups of four it leaves three CUs in four idle while the four waves of a CU share its address unit and LDS:
// launched as 256 workgroups of ONE wave the same step takes 7.0 instead of 8.0 us (image batch, draw records included;
// sf_launch_step picks the smallest BLK that still fills every CU).
// BLKP = 1000 + BLK is a SPLIT launch: BLK envs per workgroup and a second wave per tile that moves the tile's missile pool
// while the first plays the 64 games (see "the tile's MISSILE wave" below); what sf_launch_step takes for the plain step of the
// default observation of batches up to 65 536 envs (beyond, a SIMD has several games' waves to interleave anyway).
template <bool AUTOTURN, bool SHAPED, bool FUSED, int OBSK, bool XTRA, int BLKP>
__global__ __launch_bounds__(BLKP > 1000 ? 2 * (BLKP - 1000) : BLKP) void sf_step_kernel(unsigned char* state_p, const double* consts_p,
                                                          const void* actions, int n_envs_p, int act_type,
                                                          int32_t* reward_out, uint8_t* done_out, uint8_t* info_out,
                                                          SfKernelArgs a, void* obs, int obs_vec_ok, int n_steps) {
  // The five leading parameters (= a.state, a.consts, the actions, a.n_envs, the action type) are what round trip 1
  // needs; the library is built with -amdgpu-kernarg-preload-count=8, so the command processor hands them over in
  // SGPRs at wave launch and the state loads issue without a scalar-load round trip to the kernel-argument
  // segment first (on a firmware without the feature the compiler's compatibility preamble loads them).  The
  // three output pointers ride along: the epilogue then stores without a scalar load and its wait.
  // BLKP = 1000 + BLK: a SPLIT launch -- BLK envs per workgroup and as many threads again, the tiles' MISSILE waves (below)
  constexpr bool SPLIT = BLKP > 1000;
  constexpr int BLK = SPLIT ? BLKP - 1000 : BLKP;  // envs per workgroup
  constexpr int NT = SPLIT ? 2 * BLK : BLK;        // threads per workgroup
  static_assert(!SPLIT || (!FUSED && OBSK == 1 && !XTRA), "the split launch exists for the plain step of the default observation");
  constexpr int kTrigPieces = (SF_LDS_DOUBLES / 2 + NT - 1) / NT;
  // (the LDS map above, for BLK envs; a split launch's hand-over words sit between the atan table and the staging rows:
  //  the new missiles' (x, y) [BLK d2_t] and meta words [BLK], then four words per tile: fired, pool done | count, stores done)
  constexpr int kLdsAtab = SF_LDS_DOUBLES + BLK, kLdsHand = kLdsAtab + SF_ATAB_DOUBLES, kLdsHandMeta = kLdsHand + 2 * BLK,
                kLdsHandFlags = kLdsHandMeta + BLK / 2, kLdsStage = SPLIT ? kLdsHandFlags + BLK / 32 : kLdsHand;
  extern __shared__ double lds[];  // [SF_LDS_DOUBLES] cos/sin table, the event words, then the obs staging rows (SF_LDS_*)
  const unsigned tid_all = threadIdx.x;
  const unsigned tid = SPLIT ? (tid_all & (unsigned)(BLK - 1)) : tid_all;  // the env of the workgroup this thread works for
  const unsigned i = blockIdx.x * BLK + tid;  // env index: actions and outputs
  const unsigned lane = tid & 63u;
  // this wave's tile: wave-uniform by construction, made scalar for the compiler
  unsigned char* const tb = state_p + (size_t)__builtin_amdgcn_readfirstlane(i >> 6) * sfl::kTileBytes;
  const Off o = {lane * 16u, lane * 8u, lane * 4u, lane * 2u, lane};  // lane offsets inside the tile's rows
  const Off g = {i * 16u, i * 8u, i * 4u, i * 2u, i};     // env offsets into the caller's arrays
  const bool real = i < (unsigned)n_envs_p;  // lanes in [n_envs, lanes) are padding: they run NOOPs
  // predicated access to one projectile slot of the lane: `goff` = SF_GOFF(group, slot), wave-uniform
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tb, 0, (int)sfl::kTileBytes, 0x00020000);
  constexpr int kStAux = SF_STORE_MODE == 2 ? SF_SC_AUX : 0;
  auto pld16 = [&](unsigned goff, bool p) __attribute__((always_inline)) -> d2_t {
    return __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, p ? o.o16 : SF_OOB, goff, 0));
  };
  auto pst16 = [&](unsigned goff, bool p, d2_t v) __attribute__((always_inline)) {
    sf_buf_st128<kStAux>(__builtin_bit_cast(u4_t, v), rs, p ? o.o16 : SF_OOB, goff);
  };
  // the same with a per-lane slot offset (`extra` = slot * row bytes)
  auto pst16_at = [&](unsigned goff, bool p, unsigned extra, d2_t v) __attribute__((always_inline)) {
    sf_buf_st128<kStAux>(__builtin_bit_cast(u4_t, v), rs, p ? o.o16 + extra : SF_OOB, goff);
  };
#ifdef SF_STAMPS
  unsigned long long stamp_[16] = {};
  stamp_[12] = __builtin_amdgcn_s_memrealtime();
#endif
  SF_STAMP(0, false);

#ifndef SF_STAGGER
#define SF_STAGGER 12 /* x 64 cycles; A/B at 65 536 envs: 0: 7.02 us per launch, 8: 6.84, 16: 6.84, 24: 6.89, 40: 7.36 */
#endif
#if SF_STAGGER
  // Half of the workgroups (every other one of an XCD's: workgroup i runs on XCD i % 8) start a third of a microsecond
  // late: a launch is a chip-wide load burst, then arithmetic with the fabric idle, then a store burst; staggered, one
  // half's bursts meet the other half's arithmetic.
  // (a batch that leaves CUs idle has no burst to split; the batch size is a preloaded kernel argument, gridDim is a load)
#ifndef SF_STAGGER_PHASES
#define SF_STAGGER_PHASES 2 /* A/B (tools/ab.py, round 4): 3 or 4 phases of SF_STAGGER x 64 cycles each, see NOTES.md */
#endif
#if SF_STAGGER_PHASES == 2
  if (n_envs_p > 65536 - 256 && ((blockIdx.x >> 3) & 1u)) __builtin_amdgcn_s_sleep(SF_STAGGER);
#else
  if (n_envs_p > 65536 - 256) {
    const unsigned ph = (blockIdx.x >> 3) % SF_STAGGER_PHASES;  // uniform
    if (ph == 1) __builtin_amdgcn_s_sleep(SF_STAGGER);
    if (ph == 2) __builtin_amdgcn_s_sleep(2 * SF_STAGGER);
    if (ph == 3) __builtin_amdgcn_s_sleep(3 * SF_STAGGER);
  }
#endif
#endif
  // A poll of a hand-over word gives up after 2^18 rounds of 64+ cycles (10 ms; a launch lasts 6 us): a kernel must end
  // whatever happens to the other wave.  Giving up is counted (SF_ACC_HANDOVER -> sf_check_state fails: the step is wrong).
  constexpr unsigned kSpinLimit = 1u << 18;
  typedef volatile __attribute__((address_space(3))) unsigned lds_word_t;  // (named as LDS: a volatile access through a generic pointer is a flat_load)
  lds_word_t* const hflags = (lds_word_t*)(reinterpret_cast<unsigned*>(lds + kLdsHandFlags) + 4u * (tid >> 6));  // (split launches: this tile's hand-over words)
  if constexpr (SPLIT) {
    // ================= the tile's MISSILE wave =================
    // At the metric's batch a wave is alone on its SIMD and everything it does is one dependent chain; the missile pool --
    // move, test, compact 64 entries a row, whoever owns them -- depends on the 64 games only through the missiles fired
    // this tick, and the games depend on it only through the owners' event words.  A split launch gives every tile a
    // second wave (threads BLK .. 2 BLK - 1 of the workgroup: wave w + BLK / 64 works for wave w's tile; with 256 envs per
    // workgroup on the same SIMD, with fewer on one of the CU's idle ones) that does
    // exactly that part under the first wave's ship / fortress / shell arithmetic.  Hand-over through LDS, both ways by a
    // word the other side polls (a workgroup barrier would make the games wait for the pool's loads): the games' wave
    // files its new missiles and sets `fired`; the missile wave sets `done | entries kept` behind its last event word.
    // Both waves of a workgroup are resident before either starts, each sets its word on every path, the polls sleep.
    // What it buys is the first wave's issue bubbles, not the missiles' whole cost (removing them: -0.68 us; moving them
    // here: -0.12): in its arithmetic the first wave keeps the SIMD's one VALU busy most of the time, and the second wave's
    // instructions take the same issue slots.  Issue priority for the first wave and a later start for the pool's loads
    // changed nothing measurable; starting them 1 500 cycles later made the games wait; asked for without waiting for the
    // pool's count (SF_SPLIT_UNCOND): nothing either (tools/ab.py, NOTES.md, profiles/r04_split_ab.txt).
    if (tid_all >= (unsigned)BLK) {  // wave-uniform
#ifndef SF_SPLIT_PPRIO
#define SF_SPLIT_PPRIO 0 /* A/B: the missile wave at this issue priority (s_setprio): the tiles with the fullest pools are the launch's last */
#endif
#if SF_SPLIT_PPRIO
      __builtin_amdgcn_s_setprio(SF_SPLIT_PPRIO);
#endif
      // the pool's count rides in every lane's misc chunk: lane 0's word, by a scalar load
      const unsigned n_word = *reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(
          reinterpret_cast<const __attribute__((address_space(4))) void*>(
              (unsigned long long)(tb + (unsigned)sfl::chunk_offset(SF_G_misc, 0) + 8u)));
      unsigned cpi0[kTrigPieces];  // (this wave's share of the cos/sin table's pieces: one with 512 threads, up to three)
      d2_t cst0[kTrigPieces];
#pragma unroll
      for (int k = 0; k < kTrigPieces; k++) {
        cpi0[k] = min(tid_all + k * NT, (unsigned)(SF_LDS_DOUBLES / 2 - 1));
        cst0[k] = SF_LD(d2_t, (const unsigned char*)consts_p, cpi0[k] * 16u);
      }
      const unsigned m_live = n_word >> SF_MPOOL_SHIFT;
      d2_t prow[SF_MROWS];
      unsigned pmeta[SF_MROWS];
#ifndef SF_SPLIT_LATE
#define SF_SPLIT_LATE 1 /* the pool's rows are asked for once the table piece is in, i.e. behind the launch's first burst */
#endif
#if SF_SPLIT_LATE
#pragma unroll
      for (int k = 0; k < kTrigPieces; k++) reinterpret_cast<d2_t*>(lds)[cpi0[k]] = cst0[k];
      if (lane < 4u) hflags[lane] = 0u;
#if SF_SPLIT_LATE > 1
      __builtin_amdgcn_s_sleep(SF_SPLIT_LATE);
#endif
#endif
#pragma unroll
      for (int r = 0; r < SF_MROWS; r++) {
#ifndef SF_SPLIT_UNCOND
#define SF_SPLIT_UNCOND 0 /* A/B: the rows asked for whatever the pool's count is (no wait for the count in front of them) */
#endif
        const bool in_ = SF_SPLIT_UNCOND || 64u * r + lane < m_live;
        prow[r] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, in_ ? o.o16 : SF_OOB, SF_GOFF(missile_pos, r), 0));
        pmeta[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, in_ ? o.o4 : SF_OOB, SF_GOFF(missile_meta, r), 0);
      }
#if !SF_SPLIT_LATE
#pragma unroll
      for (int k = 0; k < kTrigPieces; k++) reinterpret_cast<d2_t*>(lds)[cpi0[k]] = cst0[k];
      if (lane < 4u) hflags[lane] = 0u;
#endif
      __syncthreads();  // (the workgroup's one barrier: the cos/sin table is in LDS, the hand-over words are zero)
      const double* trig = lds;
      d2_t pcs[SF_MROWS];
#pragma unroll
      for (int r = 0; r < SF_MROWS; r++) pcs[r] = *reinterpret_cast<const d2_t*>(&trig[2 * SF_MM_ANGLE(pmeta[r])]);
      double kv_speed = sfc::missile_speed, kv_fx = sfc::fort_x, kv_fy = sfc::fort_y, kv_w = sfc::width_d, kv_h = sfc::height_d,
             kv_mr2 = sfc::missile_hit_r2;
      asm volatile("" : "+v"(kv_speed), "+v"(kv_fx), "+v"(kv_fy), "+v"(kv_w), "+v"(kv_h), "+v"(kv_mr2));
      unsigned wp = 0;
      unsigned* const evw32 = reinterpret_cast<unsigned*>(lds + SF_LDS_EV) + 2u * (tid & ~63u);
      auto m_row = [&](double x, double y, unsigned meta, d2_t cs, bool valid) __attribute__((always_inline)) {  // (as below)
        const double nx = x + kv_speed * cs.x, ny = y + kv_speed * cs.y;
        const double dx = nx - kv_fx, dy = ny - kv_fy;
        const bool hit = valid & (dx * dx + dy * dy <= kv_mr2);
        const bool gone = valid & (hit | (__builtin_fmax(__builtin_fmax(-nx, nx - kv_w), __builtin_fmax(-ny, ny - kv_h)) > 0));
        if (__ballot(gone) != 0ull) {
          if (gone)
            __hip_atomic_fetch_or(evw32 + 2 * SF_MM_OWNER(meta) + (hit ? 0 : 1), 1u << SF_MM_SLOT(meta), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const bool keep = valid & !gone;
        const unsigned long long kb = __ballot(keep);
        const unsigned idx = wp + __builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u));
        wp += (unsigned)__popcll(kb);
        sf_buf_st128<kStAux>(__builtin_bit_cast(u4_t, (d2_t{nx, ny})), rs, keep ? idx * 16u : SF_OOB,
                                               SF_GOFF(missile_pos, 0));
        __builtin_amdgcn_raw_buffer_store_b32(meta, rs, keep ? idx * 4u : SF_OOB, SF_GOFF(missile_meta, 0), kStAux);
      };
#pragma unroll
      for (int r = 0; r < SF_MROWS; r++)
        if (m_live > 64u * r) m_row(prow[r].x, prow[r].y, pmeta[r], pcs[r], 64u * r + lane < m_live);
      if (m_live > 64u * SF_MROWS) {
#pragma unroll 1
        for (unsigned r = SF_MROWS; 64u * r < m_live; r++) {
          const d2_t pr = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, o.o16 + r * 1024u, SF_GOFF(missile_pos, 0), 16));
          const unsigned pm = __builtin_amdgcn_raw_buffer_load_b32(rs, o.o4 + r * 256u, SF_GOFF(missile_meta, 0), 16);
          m_row(pr.x, pr.y, pm, *reinterpret_cast<const d2_t*>(&trig[2 * SF_MM_ANGLE(pm)]), 64u * r + lane < m_live);
        }
      }
      // the missiles fired this tick (SRC/game.cpp:237-238), one more row, lane = owner
      unsigned fired, spins = 0u;
      while ((fired = (unsigned)__builtin_amdgcn_readfirstlane((int)hflags[0])) == 0u && ++spins < kSpinLimit) __builtin_amdgcn_s_sleep(1);
      if (spins >= kSpinLimit && lane == 0u) atomicAdd(&a.acc[SF_ACC_HANDOVER], 1ull);  // (never seen; see kSpinLimit)
      asm volatile("" ::: "memory");
      if (fired == 2u) {
        const d2_t nxy = reinterpret_cast<const d2_t*>(lds + kLdsHand)[tid];
        const unsigned nm = reinterpret_cast<const unsigned*>(lds + kLdsHandMeta)[tid];
        m_row(nxy.x, nxy.y, nm & 0x7FFFFFFFu, *reinterpret_cast<const d2_t*>(&trig[2 * SF_MM_ANGLE(nm)]), !(nm >> 31));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the event words are in
      if (lane == 0u) hflags[1] = 0x80000000u | wp;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pool's rows are written (the rare purge below re-reads them)
      if (lane == 0u) hflags[2] = 1u;
      return;
    }
  }
#ifndef SF_SPLIT_PRIO
#define SF_SPLIT_PRIO 0 /* A/B: the games' wave at this issue priority over its missile wave (s_setprio) */
#endif
#if SF_SPLIT_PRIO
  if constexpr (SPLIT) __builtin_amdgcn_s_setprio(SF_SPLIT_PRIO);
#endif
  // ================= round trip 1: every unconditional load =================
  // act_type SF_ACT_SAMPLED: no action array -- `actions` is this batch's sampler records (SfActRec, one per tile) and the
  // lane draws its action itself: Philox4x32-10 keyed by the seed, counter (lane of the whole job, tick).  The record is
  // ONE scalar load off a preloaded kernel argument, back long before the early set, and the ten rounds run while the
  // wave would otherwise sit in that wait.  (constant address space = s_load; the tile's tick is rewritten by lane 0 after
  // the last tick of the launch, nothing reads it again in between.)
  u4_t act_rec = {0u, 0u, 0u, 0u};
  if (XTRA && act_type == SF_ACT_SAMPLED)
    act_rec = *(reinterpret_cast<const __attribute__((address_space(4))) u4_t*>(
                    reinterpret_cast<const __attribute__((address_space(4))) void*>((unsigned long long)actions)) +
                __builtin_amdgcn_readfirstlane(i >> 6));
  // (its own variable, never the destination of a load: joined with the loaded action in one register, the compiler
  //  waits for every outstanding load -- vmcnt(0) -- in front of the ten rounds instead of running them under that wait)
  auto sample_action = [&](int step) __attribute__((always_inline)) -> int {
    const unsigned x = sf_philox4x32_10(act_rec.w + lane, act_rec.x + (unsigned)step, act_rec.y, act_rec.z);
    return real ? (int)__umulhi(x, (unsigned)a.n_actions) : 0;
  };
  auto load_action = [&](int step) __attribute__((always_inline)) -> int {  // ENV:211-212
    if (!real || (XTRA && act_type == SF_ACT_SAMPLED)) return 0;
    const unsigned char* ab = (const unsigned char*)actions;
    const unsigned e = (unsigned)step * (unsigned)n_envs_p + i;
    if (act_type == 8) return (int)SF_LD(long long, ab, e * 8u);
    if (act_type == 4) return SF_LD(int, ab, e * 4u);
    return SF_LD(unsigned char, ab, e);
  };
  // ---- the early set: what keys, respawn and ship need (load_lane_early), the action, the cos/sin table pieces
  Lane L;
  load_lane_early(tb, o, L);  // before the action: its address needs two more kernel arguments and a branch on the action type
  int act_next = load_action(0);
  const unsigned atab_pi = min(tid, (unsigned)(SF_ATAB_DOUBLES / 2 - 1));
  const d2_t atab_piece = SF_LD(d2_t, (const unsigned char*)consts_p, (SF_CONST_ATAB / 2 + atab_pi) * 16u);
  // cos/sin table: 720 doubles = 360 16-byte pieces, six per lane (the last one partial)
  const unsigned char* cb = (const unsigned char*)consts_p;
  // (threads past the end re-load and re-store the last piece: straight-line code, no exec-masked
  // branches for the compiler's wait-count insertion to be conservative about)
  d2_t cst[kTrigPieces];
  unsigned cpi[kTrigPieces];
#pragma unroll
  for (int k = 0; k < kTrigPieces; k++) {
    cpi[k] = min(tid_all + k * NT, (unsigned)(SF_LDS_DOUBLES / 2 - 1));
    cst[k] = SF_LD(d2_t, cb, cpi[k] * 16u);
  }
  // (behind the last load of the early set: the ten rounds run while those are in flight)
  int act_sampled = 0;
  if (XTRA && act_type == SF_ACT_SAMPLED) act_sampled = sample_action(0);  // uniform branch, VALU only
  // lane l takes entry 64 r + l if the pool has that many (`n_pool`: the tile's count, known once the early set is in;
  // ~0u = not known yet, take everything): the instructions are unconditional, the bytes are not
#define SF_LOAD_POOL_ROWS(aux, n_pool)                                                                                        \
  _Pragma("unroll") for (int r = 0; r < SF_MROWS; r++) {                                                                      \
    const bool in_ = 64u * r + lane < (n_pool);                                                                               \
    prow[r] = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, in_ ? o.o16 : SF_OOB, SF_GOFF(missile_pos, r), (aux)));   \
    pmeta[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, in_ ? o.o4 : SF_OOB, SF_GOFF(missile_meta, r), (aux));               \
  }
  constexpr bool kPoolLoads = !SPLIT && (SF_ABL_PROJ == 0 || SF_ABL_PROJ == 3);
  d2_t prow[SF_MROWS];
  unsigned pmeta[SF_MROWS];
  LaneLate late;
#ifndef SF_LATE
#define SF_LATE 1 /* 0: the late set rides with the early one (A/B) */
#endif
#if !SF_LATE
  late = load_lane_late(tb, o);
  if (kPoolLoads) { SF_LOAD_POOL_ROWS(0, ~0u) }
#endif
  SF_STAMP(1, false);
  SF_STAMP(2, true);

  // cos/sin table -> LDS (the loads were issued in round trip 1).  BEFORE the projectile loads are issued:
  // memory returns in order and the number of predicated loads below is not known at compile time, so a
  // wait for the table placed after them is a wait for all of them (`s_waitcnt vmcnt(0)`) -- round trip 2
  // would be over before the workgroup barrier instead of running under the ship / fortress arithmetic.
  // For the same reason EVERY load issued so far is waited for here, explicitly (vmcnt(0), the other counters
  // untouched): a first use of, say, the flags after the predicated loads would otherwise be a vmcnt(0) too.
  __builtin_amdgcn_s_waitcnt(0x0F70);
#if SF_ABL_TRIG != 2
#pragma unroll
  for (int k = 0; k < kTrigPieces; k++) reinterpret_cast<d2_t*>(lds)[cpi[k]] = cst[k];
#endif
  // this lane's (hit, out) event words start at zero; the missile pool's entries OR their slot bit into their
  // OWNER's words (wave-private: only lanes of this wave own entries of this tile; LDS is in order per wave)
  unsigned long long* const evw = reinterpret_cast<unsigned long long*>(lds + SF_LDS_EV) + (tid & ~63u);
  evw[lane] = 0ull;
  reinterpret_cast<d2_t*>(lds + kLdsAtab)[atab_pi] = atab_piece;  // (every lane the same nine pieces: no branch)

  // ================= round trip 2: live shell slots, predicated by the alive mask ======
  // Slot groups (pairs): a wave ballot skips a group no lane uses.  The kernel lasts as long as its slowest wave, so
  // the groups reach well past the common case: the dependent-load loop behind them is for slots hardly ever used.
  // (Missiles need no second round trip: the tile's pool rows came with the first.)
  double shx[SF_SPF], shy[SF_SPF], shvx[SF_SPF], shvy[SF_SPF];
#ifndef SF_SHELL_INIT
#define SF_SHELL_INIT 0
#endif
#pragma unroll
  for (int s = 0; s < SF_SPF; s++) {
    if (FUSED || SF_SHELL_INIT) {
      shx[s] = shy[s] = shvx[s] = shvy[s] = 0;
    } else {
      // one tick per launch: a slot's registers are read only under the same wave-wide test that loaded them, and what a
      // lane without a shell in the slot computes from them is masked out (`live`): whatever the registers hold will do
      asm volatile("" : "=v"(shx[s]), "=v"(shy[s]), "=v"(shvx[s]), "=v"(shvy[s]));
    }
  }
  // a dead ship whose explosion is over respawns this tick (SRC/game.cpp:151-157): fetch its
  // entry of the spawn sequence now, not in the middle of the arithmetic
  bool will_respawn = !(L.fl & SF_FL_SHIP_ALIVE) && L.death_t >= sfc::explode_duration;
  unsigned long long spawn_e = 0;
  // The spawn entry FIRST and without a branch (a lane that does not respawn gets an out-of-range offset): memory
  // returns in order, so what is issued behind it -- shells, the late set -- is not waited for when the respawn reads it
  const __amdgpu_buffer_rsrc_t rs_spawn =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(a.spawn), 0, (int)((a.spawn_mask + 1u) * 8u), 0x00020000);
  auto load_spawn = [&]() __attribute__((always_inline)) {
#if SF_ABL_SPAWN  /* timing-only: no dependent spawn-entry load (WRONG results) */
    spawn_e = 0x0000005A00C800C8ull + (L.cursor & 63u);
#else
    const i2_t e = __builtin_bit_cast(i2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_spawn, will_respawn ? (L.cursor & a.spawn_mask) * 8u : SF_OOB, 0, 0));
    spawn_e = (unsigned long long)(unsigned)e.x | ((unsigned long long)(unsigned)e.y << 32);
#endif
  };
  load_spawn();
  {
#pragma unroll
    for (int g = 0; g < SF_SPF / SF_SGSZ; g++) {
      if (SF_ABL_PROJ < 2 && __ballot((L.smask & (((1u << SF_SGSZ) - 1u) << (SF_SGSZ * g))) != 0u) != 0ull) {
#pragma unroll
        for (int k = 0; k < SF_SGSZ; k++) {
          const int s = SF_SGSZ * g + k;
          const bool live = (L.smask >> s) & 1u;
          const d2_t sp = pld16(SF_GOFF(shell_pos, s), live);
          const d2_t sv = pld16(SF_GOFF(shell_vel, s), live);
          shx[s] = sp.x;
          shy[s] = sp.y;
          shvx[s] = sv.x;
          shvy[s] = sv.y;
        }
      }
    }
  }
#if SF_LATE
  // ---- the late set, 2 + 2 SF_MROWS unconditional memory instructions behind everything the first phases wait for: the
  //      score and counts chunks, then the first SF_MROWS rows of the tile's missile pool -- lane l takes entry 64 r + l,
  //      (x, y) and the meta word, if the pool has that many (the count came with the early set): unconditional
  //      instructions, so the compiler can count them when it waits for the spawn entry issued ahead of them
  late = load_lane_late(tb, o);
  if (kPoolLoads) { SF_LOAD_POOL_ROWS(0, L.mpool) }
#endif

#if SF_ABL_TRIG == 0
  __syncthreads();  // the only workgroup barrier of the kernel
#endif
  const double* trig = lds;
  SF_STAMP(3, false);

  unpack_lane_late(late, L);  // names only: the wait for the late set sits at the first real use
  const int n_iter = FUSED ? n_steps : 1;
  for (int step = 0; step < n_iter; step++) {
  int act = (XTRA && act_type == SF_ACT_SAMPLED) ? act_sampled : act_next;
  if (FUSED) {
    if (step + 1 < n_iter) {
      act_next = load_action(step + 1);  // in flight while this tick computes
      if (XTRA && act_type == SF_ACT_SAMPLED) act_sampled = sample_action(step + 1);
    }
    if (step > 0) {
      // the pool rows the previous tick compacted in place: agent-scope loads (they bypass the wave's L1, where the rows
      // read a tick ago may still sit), in flight under the key / ship / fortress / shell arithmetic
      if (SF_ABL_PROJ == 0 || SF_ABL_PROJ == 3) { SF_LOAD_POOL_ROWS(16, L.mpool) }
      will_respawn = !(L.fl & SF_FL_SHIP_ALIVE) && L.death_t >= sfc::explode_duration;
      load_spawn();
    }
  }
  const size_t so = (size_t)step * (size_t)a.n_envs;  // this tick's row of the output arrays
  // image batches: this tick leaves the envs' draw records for the frame kernel (sf_drawrec.h) -- of a fused launch, the
  // last tick does.  `dr_proj`: where this env's projectiles come near the score / the bar (sfd::hud_flags_near)
  const bool draw_now = OBSK != 1 && a.draw != nullptr && (!FUSED || step + 1 == n_iter);  // uniform
  unsigned dr_proj = 0u;

  const int act_raw = act;  // what rollouts.actions[step] records
  if (act < 0 || act >= a.n_actions) {
    atomicAdd(&a.acc[SF_ACC_BAD_ACTION], 1ull);  // reference: IndexError; here NOOP + counted (sf_check_actions)
    act = 0;
  }
  const unsigned keys = (unsigned)(a.action_keys >> (4 * act)) & 0xFu;

  StatDelta S;

  // ================= Game::stepOneTick (SRC/game.cpp:473-485) =================
  float rew = 0;        // mReward = 0 (updateTime: with the game-over test below, `time` is in a late chunk)

  // ---- processKeyState (SRC/game.cpp:218-272); the wrapper sends FIRE, THRUST, (LEFT, RIGHT)
  //      press-or-release every step (ENV:213-229), so each key is one edge test.  Branch-free:
  //      the flag bits of `fl` sit in key order (FIRE, THRUST, LEFT, RIGHT = bits 2..5), so the
  //      new flags are the key mask itself, an edge is key XOR flag, a press edge is key AND NOT
  //      flag; an edge zeroes that key's timer (:227,231,235,244,250,253,257,261), a press edge
  //      counts (:228,232,236,245).
  int new_m_slot = -1;
  unsigned key_edge_bits = 0;
  float amt_fire = 0.0f, amt_hex = 0.0f;  // this tick's first two score() amounts, applied in order once `points` is needed
  unsigned key_edges;  // SF_EV_PRESS_* | SF_EV_RELEASE_*: the key STATE changes of this tick
  {
    const unsigned kmask = AUTOTURN ? 0x3u : 0xFu;  // autoturn games send two keys (ENV:221)
    const unsigned old = (L.fl >> 2) & kmask, now = keys & kmask;
    const unsigned edge = old ^ now, press = now & ~old;
    key_edges = press | ((old & ~now) << 4);
    L.fl = (L.fl & ~(kmask << 2)) | (now << 2);
    key_edge_bits = edge;  // the key timers live in late chunks: an edge zeroes its timer at stepTimers below
    S.shots += (int)(press & 1u);
    S.thrusts += (int)((press >> 1) & 1u);
    L.kc0 += (press & 1u) | ((press & 2u) << 15);  // shots in the low half, thrusts in the high half
    if (!AUTOTURN) L.kc1 += ((press >> 2) & 1u) | ((press & 8u) << 13);  // lefts, rights
    if (!AUTOTURN) {
      S.lefts += (int)((press >> 2) & 1u);
      S.rights += (int)((press >> 3) & 1u);
    }
    // FIRE press edge: fireMissile (SRC/game.cpp:175-192,237-238) before this tick's turn and
    // move; the shot is counted and the timer zeroed even if nothing could be created
    // as selects: an exec-mask branch around a three-line body costs a lone wave more than the body (the
    // compare travels VALU -> SALU -> EXEC -> VALU).  score(+0.0f) changes nothing: x + 0.0f == x bit for bit for
    // every x these sums can hold (none of them is ever -0.0f), and the clamp does not move a non-negative total.
    {
      const int slot = __ffs(~L.mmask) - 1;
      const bool can = (press & 1u) && (L.fl & SF_FL_SHIP_ALIVE) && slot >= 0 && slot < SF_NSLOT;
      new_m_slot = can ? slot : -1;
      L.mmask |= can ? (1u << (slot & 31)) : 0u;
      amt_fire = can ? -sfc::Score<SHAPED>::missile_penalty : 0.0f;  // reward(-penalty), :187
    }
  }
  // the missile created above starts at the ship's pre-move position and heading
  const double new_m_x = L.sx, new_m_y = L.sy;
  const int new_m_angle = L.angle;
  if constexpr (SPLIT) {  // ... and goes to the tile's missile wave (LDS is in order per wave: the word last)
    reinterpret_cast<d2_t*>(lds + kLdsHand)[tid] = d2_t{new_m_x, new_m_y};
    reinterpret_cast<unsigned*>(lds + kLdsHandMeta)[tid] =
        new_m_slot >= 0 ? SF_MM_PACK(new_m_angle, lane, new_m_slot & 31) : 0x80000000u;
    asm volatile("" ::: "memory");
    const unsigned fired = __ballot(new_m_slot >= 0) != 0ull ? 2u : 1u;
    if (lane == 0u) hflags[0] = fired;
  }

  // ---- monitorShipRespawn (SRC/game.cpp:151-157)
  if (will_respawn) {  // !alive && deathTimer >= shipExplodeDuration, evaluated (and fetched) above
    spawn_ship_from(a, L, spawn_e);  // (mFortress.mTimer = 0, :155: at the fortress update below, a late chunk)
  }

  // ---- updateShip (SRC/game.cpp:314-351)
  if (L.fl & SF_FL_SHIP_ALIVE) {
    if (AUTOTURN) {
      // stdAngle(ceil(angleTo(ship, fortress)))  (SRC/vector.cpp:42-52)
      double t = sf_atan2<true>(sfc::fort_y - L.sy, sfc::fort_x - L.sx, lds + kLdsAtab);
      if (t < 0) t += M_PI * 2;
      double c = ceil(rad2deg(t));  // in [0, 360]
      int ia = (int)c;
      if (ia >= 360) ia -= 360;  // fmod(360, 360)
      L.angle = ia;
    } else {
      // TURN_LEFT / TURN_RIGHT / both or none (:317-325): right - left steps of turnSpeed, then stdAngle's wrap
      // (one of the two corrections can apply, and only on the side that was stepped towards)
      L.angle += sfc::turn_speed * ((int)((L.fl / SF_FL_RIGHT) & 1u) - (int)((L.fl / SF_FL_LEFT) & 1u));
      L.angle += L.angle < 0 ? 360 : 0;
      L.angle -= L.angle >= 360 ? 360 : 0;
    }
    {
      const bool th = L.fl & SF_FL_THRUST;
      const double tvx = L.vx + sfc::ship_accel * SF_COS(L.angle), tvy = L.vy + sfc::ship_accel * SF_SIN(L.angle);
      L.vx = th ? tvx : L.vx;
      L.vy = th ? tvy : L.vy;
    }
    L.sx += L.vx;
    L.sy += L.vy;
    // `if (!big.isInside) ... else if (small.isInside) ...` (:337-349), counters branch-free
    const int out_big = !inside_big_hex(L.sx, L.sy);
    const int in_small = !out_big & inside_small_hex(L.sx, L.sy);
    {
      const int dead_now = out_big | in_small;  // killShip (:274-280) on a live ship, as selects
      L.fl &= dead_now ? ~SF_FL_SHIP_ALIVE : ~0u;
      L.death_t = dead_now ? 0 : L.death_t;
      S.ship_deaths += dead_now;
      amt_hex = dead_now ? -sfc::Score<SHAPED>::death_penalty : 0.0f;  // penalize(shipDeathPenalty), :339,345
    }
    S.big_hex_deaths += out_big;
    S.small_hex_deaths += in_small;
  }

  // the two bearings the rest of the tick and the observation need, side by side (ILP)
#if SF_ABL_ATAN == 2
  double a_pos = (L.sy - sfc::fort_y) * 0.001 + (L.sx - sfc::fort_x) * 0.002;
#else
  double a_pos = sf_atan2<true>(L.sy - sfc::fort_y, L.sx - sfc::fort_x, lds + kLdsAtab);
#endif
#if SF_ABL_ATAN
  double a_vel = L.vy * 0.5 + L.vx;
#elif SF_AXIS_VEL
  double a_vel = sf_atan2<false>(L.vy, L.vx);
#else
  double a_vel = SF_FAST_ATAN ? sf_atan2_core(L.vy, L.vx, lds + kLdsAtab) : atan2(L.vy, L.vx);
#endif

  // ---- updateFortress (SRC/game.cpp:194-216)
  int new_s_slot = -1;
  double new_s_vx = 0, new_s_vy = 0;
  bool fort_respawned = false;
  {
    double ats = rad2deg(a_pos);  // stdAngle: in [-180, 180], only the sign fix applies
    if (ats < 0) ats += 360;
    L.fort_t = will_respawn ? 0 : L.fort_t;  // monitorShipRespawn's mFortress.mTimer = 0 (SRC/game.cpp:155)
    fort_respawned = !(L.fl & SF_FL_FORT_ALIVE) && L.fort_death_t > sfc::fort_respawn;
    L.fort_t = fort_respawned ? 0 : L.fort_t;
    L.fl |= fort_respawned ? SF_FL_FORT_ALIVE : 0u;
    const bool ship_up = L.fl & SF_FL_SHIP_ALIVE;
    {
      double q = ceil(SF_DIV(ats, sfc::sector_size)) * sfc::sector_size;  // in [0, 360]
      int fa = (int)q;
      fa -= fa >= 360 ? 360 : 0;
      L.fort_angle = ship_up ? fa : L.fort_angle;
      const bool moved = ship_up && fa != L.fort_last;
      L.fort_last = moved ? fa : L.fort_last;
      L.fort_t = moved ? 0 : L.fort_t;
    }
    if (ship_up) {
      if (L.fort_t >= sfc::lock_time && (L.fl & SF_FL_FORT_ALIVE)) {
        // fireShell (SRC/game.cpp:159-173): vel = shellSpeed * (cos, sin)(deg2rad(angle_to_ship)).
        // angle_to_ship is the bearing of d = ship - fortress, so (cos, sin) = d / |d|: one sqrt
        // and two divisions instead of a device sincos with argument reduction.  Shell velocity
        // was never bit-exact against glibc's sin/cos anyway; both forms sit within a few 1e-15
        // of the true direction (shell positions are tested to 1e-9).
        int slot = __ffs(~L.smask) - 1;
        if (slot < SF_NSLOT) {
          new_s_slot = slot;
          L.smask |= 1u << slot;
          const double ddx = L.sx - sfc::fort_x, ddy = L.sy - sfc::fort_y;
          const double nrm = sqrt(ddx * ddx + ddy * ddy);
          new_s_vx = sfc::shell_speed * (ddx / nrm);
          new_s_vy = sfc::shell_speed * (ddy / nrm);
        }
        L.fort_t = 0;
      }
    }
  }
  SF_STAMP(4, false);
  SF_STAMP(5, true);

  // keep the spawn entry's register allocated up to here: reused earlier, the compiler must first wait for the
  // (predicated, possibly outstanding) load into it -- a vmcnt(0), i.e. the whole of round trip 2, right
  // after the key processing
  asm volatile("" ::"v"(spawn_e));
  // the (cos, sin) of the pool rows' headings in ONE batch of LDS reads, here, so that their latency hides under the
  // shells (a wave alone on its SIMD hides nothing: looked up row by row below, each would pay its own LDS round trip)
  const unsigned m_live = (unsigned)__builtin_amdgcn_readfirstlane((int)L.mpool);  // entries in the pool before this tick
  d2_t pcs[SF_MROWS];
#pragma unroll
  for (int r = 0; r < SF_MROWS; r++)
    pcs[r] = (SPLIT || SF_ABL_PROJ == 1 || SF_ABL_PROJ == 2) ? d2_t{0, 0} : *reinterpret_cast<const d2_t*>(&trig[2 * SF_MM_ANGLE(pmeta[r])]);

  // the tick's first two score() calls (fireMissile's penalty, then a hexagon death), in their order, now that the
  // score chunk is needed anyway (shell kills and missile events follow)
  score(amt_fire, rew, L);
  score(amt_hex, rew, L);

  // The constants of the projectile arithmetic, parked in VGPRs: none of them is an inline constant, SGPRs are
  // scarce (101 of 102 in use), and the compiler re-materialises each as an `s_mov_b32` pair in front of every slot
  // group -- about a hundred scalar moves per tick that a lone wave pays in full.  Opaque to the optimiser on
  // purpose; the values are the same doubles.
  double kv_speed = sfc::missile_speed, kv_fx = sfc::fort_x, kv_fy = sfc::fort_y, kv_w = sfc::width_d, kv_h = sfc::height_d,
         kv_mr2 = sfc::missile_hit_r2, kv_sr2 = sfc::shell_hit_r2;
  asm volatile("" : "+v"(kv_speed), "+v"(kv_fx), "+v"(kv_fy), "+v"(kv_w), "+v"(kv_h), "+v"(kv_mr2), "+v"(kv_sr2));
  // Game::isOutsideGameArea (SRC/game.cpp:129-131), (x < 0) | (x > W) | (y > H) | (y < 0), as ONE comparison:
  // the largest of -x, x - W, -y, y - H is positive exactly when one of the four holds (a difference of two
  // doubles has the exact sign; no NaN on this path).  Four compares OR-ed through scalar registers are a chain of
  // VALU -> SALU -> VALU hand-offs per slot, which a wave alone on its SIMD waits out every time.
  auto outside = [&](double x, double y) __attribute__((always_inline)) -> bool {
    return __builtin_fmax(__builtin_fmax(-x, x - kv_w), __builtin_fmax(-y, y - kv_h)) > 0;
  };

  // ---- updateShells (SRC/game.cpp:404-423).  Ballistics of the prefetched slots first, as
  //      straight-line code; then the (ship-alive dependent) outcome in slot order.
  // What is written ONCE per projectile -- a new shell's velocity -- is one store for the
  // wave with the slot in the lane's offset, not a predicated store per slot: every memory instruction of the four
  // waves of a CU goes through the one address unit they share, idle lanes or not.
  static_assert(sfl::kGroups[SF_G_shell_vel].chunk * sfl::kTileLanes == 1024, "slot stride of the row below");
  if (__ballot(new_s_slot >= 0) != 0ull)
    pst16_at(SF_GOFF(shell_vel, 0), new_s_slot >= 0, (unsigned)new_s_slot * 1024u, d2_t{new_s_vx, new_s_vy});
  {
#pragma unroll
    for (int g = 0; g < SF_SPF / SF_SGSZ; g++) {
      const unsigned gmask = ((1u << SF_SGSZ) - 1u) << (SF_SGSZ * g);
      if (SF_ABL_PROJ >= 2 || __ballot((L.smask & gmask) != 0u) == 0ull) continue;
      unsigned col = 0, out = 0;
      double nx[SF_SGSZ], ny[SF_SGSZ];
#pragma unroll
      for (int k = 0; k < SF_SGSZ; k++) {
        const int s = SF_SGSZ * g + k;
        const bool isnew = (s == new_s_slot);
        const double vx = isnew ? new_s_vx : shvx[s], vy = isnew ? new_s_vy : shvy[s];
        nx[k] = (isnew ? kv_fx : shx[s]) + vx;
        ny[k] = (isnew ? kv_fy : shy[s]) + vy;
        // Object::collided (SRC/object.cpp:12-15): sqrt(dx^2+dy^2) <= r.  With a correctly
        // rounded sqrt and r an integer, RN(sqrt(s)) <= r  <=>  s <= r^2 (r^2 is exactly
        // representable and the next double above r^2 has a root that rounds above r).
        const double dx = nx[k] - L.sx, dy = ny[k] - L.sy;
        col |= (unsigned)(dx * dx + dy * dy <= kv_sr2) << s;
        out |= (unsigned)outside(nx[k], ny[k]) << s;
        if (FUSED) {  // the registers carry the shell into the next tick
          shx[s] = nx[k];
          shy[s] = ny[k];
          shvx[s] = vx;
          shvy[s] = vy;
        }
      }
      const unsigned live = L.smask & gmask;
      col &= live;
      // slot order: the first colliding shell kills a live ship; every other shell only
      // leaves by flying out (`if (alive && collided) ... else if (outside)`, :410-420)
      unsigned dead = out & live;
      {
        const int hitship = (L.fl & SF_FL_SHIP_ALIVE) && col;
        dead |= hitship ? (col & (0u - col)) : 0u;  // lowest colliding slot
        L.fl &= hitship ? ~SF_FL_SHIP_ALIVE : ~0u;  // killShip on a live ship, as selects
        L.death_t = hitship ? 0 : L.death_t;
        S.ship_deaths += hitship;
        S.shell_deaths += hitship;
        score(hitship ? -sfc::Score<SHAPED>::death_penalty : 0.0f, rew, L);
      }
      L.smask &= ~dead;
#pragma unroll
      for (int k = 0; k < SF_SGSZ; k++) {
        const int s = SF_SGSZ * g + k;
        pst16(SF_GOFF(shell_pos, s), (L.smask >> s) & 1u, d2_t{nx[k], ny[k]});
        if (draw_now)
          if (((L.smask >> s) & 1u) && sfd::hud_rows_near((float)ny[k], sfd::kShellExt))  // (all but never)
            dr_proj |= sfd::hud_flags_near((float)nx[k], (float)ny[k], sfd::kShellExt);
      }
    }
    if (SF_ABL_PROJ < 2 && __ballot((L.smask >> SF_SPF) != 0u) != 0ull) {  // rare: more than SF_SPF shells in some lane
#pragma unroll 1
      for (int s = SF_SPF; s < SF_NSLOT; s++) {
        const bool live = (L.smask >> s) & 1u;
        if (__ballot(live) == 0ull) continue;
        if (live) {
          double x, y, vx, vy;
          if (s == new_s_slot) {
            x = sfc::fort_x;
            y = sfc::fort_y;
            vx = new_s_vx;
            vy = new_s_vy;
          } else {
            const d2_t sp = ld_coherent_d2(SF_CHUNK(shell_pos, s) + o.o16);
            const d2_t sv = ld_coherent_d2(SF_CHUNK(shell_vel, s) + o.o16);
            x = sp.x;
            y = sp.y;
            vx = sv.x;
            vy = sv.y;
          }
          x += vx;
          y += vy;
          bool dead = false;
          if (L.fl & SF_FL_SHIP_ALIVE) {
            const double dx = x - L.sx, dy = y - L.sy;
            if (dx * dx + dy * dy <= sfc::shell_hit_r2) {
              dead = true;
              kill_ship(L, S);
              score(-sfc::Score<SHAPED>::death_penalty, rew, L);
              S.shell_deaths += 1;
            }
          }
          if (!dead && outside_area(a, x, y)) dead = true;
          if (dead) {
            L.smask &= ~(1u << s);
          } else {
            SF_ST(d2_t, SF_CHUNK(shell_pos, s), o.o16, (d2_t{x, y}));
            if (draw_now) dr_proj |= sfd::hud_flags_near((float)x, (float)y, sfd::kShellExt);
          }
        }
      }
    }
  }

  // ---- updateMissiles (SRC/game.cpp:353-402) over the tile's missile POOL.  The reference walks 20 slots per env; with
  //      one lane per env a wave would walk every slot ANY of its 64 envs uses with a couple of lanes busy.  The tile's
  //      live missiles are kept as one dense list instead: a row is 64 entries, one per lane, whoever owns them, every
  //      lane busy (random play: 104 missiles per tile = 2 rows instead of 8-9 slot passes).  An entry that hits the
  //      fortress or leaves the area ORs its slot bit into its OWNER lane's event words in LDS; the survivors are
  //      compacted IN PLACE (ballot + prefix count) as the rows go by -- a row is in registers before anything is
  //      written, and entries only move down.  What a hit or a miss does to the fortress / score is order dependent
  //      (slot order within an env), so the owners replay their events from the collected masks afterwards.
  //      New missiles (fired this tick, SRC/game.cpp:237-238: before the move) are one more row, lane = owner.
  unsigned hit_count = 0;  // missiles that reached the fortress this tick (alive or not)
  {
    unsigned wp = 0;  // write pointer of the compaction = entries kept so far (wave-uniform)
    unsigned* const evw32 = reinterpret_cast<unsigned*>(evw);
    // (the draw records of this wave's 64 envs: one descriptor, like the tile's)
    const __amdgpu_buffer_rsrc_t rs_draw = __builtin_amdgcn_make_buffer_rsrc(
        draw_now ? a.draw + (size_t)__builtin_amdgcn_readfirstlane(i >> 6) * (size_t)SF_DR_TILE_BYTES : (unsigned char*)nullptr, 0,
        draw_now ? SF_DR_TILE_BYTES : 0, 0x00020000);
    auto m_row = [&](double x, double y, unsigned meta, d2_t cs, bool valid) __attribute__((always_inline)) {
      // velocity = missileSpeed * (cos, sin)(deg2rad(angle)) with an integer angle: table
      const double nx = x + kv_speed * cs.x, ny = y + kv_speed * cs.y;
      const double dx = nx - kv_fx, dy = ny - kv_fy;
      const bool hit = valid & (dx * dx + dy * dy <= kv_mr2);  // collided(mFortress), see shells
      const bool gone = valid & (hit | outside(nx, ny));       // `else if (isOutsideGameArea)`: hit wins
      if (__ballot(gone) != 0ull) {
        if (gone) {  // (slot bit) -> the owner's hit word or, for a missile that left the area, its out word
          __hip_atomic_fetch_or(evw32 + 2 * SF_MM_OWNER(meta) + (hit ? 0 : 1), 1u << SF_MM_SLOT(meta), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      const bool keep = valid & !gone;
      const unsigned long long kb = __ballot(keep);
      const unsigned idx = wp + __builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u));
      wp += (unsigned)__popcll(kb);
      sf_buf_st128<kStAux>(__builtin_bit_cast(u4_t, (d2_t{nx, ny})), rs, keep ? idx * 16u : SF_OOB,
                                             SF_GOFF(missile_pos, 0));
      __builtin_amdgcn_raw_buffer_store_b32(meta, rs, keep ? idx * 4u : SF_OOB, SF_GOFF(missile_meta, 0), kStAux);
      if (draw_now) {  // uniform.  The survivor's transform goes to its OWNER's draw record, at its slot (sf_drawrec.h)
        typedef float f4_t __attribute__((ext_vector_type(4)));
        const unsigned doff = (SF_DR_PIECE_OBJ0 + SF_DR_OBJ_MISSILE0 + SF_MM_SLOT(meta)) * (unsigned)SF_DR_PIECE_STRIDE +
                              SF_MM_OWNER(meta) * (unsigned)SF_DR_LANE_STRIDE;
        sf_buf_st128<SF_DR_AUX>(__builtin_bit_cast(u4_t, (d2_t{nx, ny})), rs_draw, keep ? doff : SF_OOB, 0);
        __builtin_amdgcn_raw_buffer_store_b16((short)SF_MM_ANGLE(meta), rs_draw,
                                              keep ? SF_DR_ANGLES_OFF + 2u * SF_MM_SLOT(meta) + SF_MM_OWNER(meta) * (unsigned)SF_DR_LANE_STRIDE : SF_OOB,
                                              0, SF_DR_AUX);
        const bool rows = keep & sfd::hud_rows_near((float)ny, sfd::kMissileExt);
        if (__ballot(rows) != 0ull) {  // (all but never) -> bits 24..27 of the owner's hit word
          if (rows)
            __hip_atomic_fetch_or(evw32 + 2 * SF_MM_OWNER(meta), sfd::hud_flags_near((float)nx, (float)ny, sfd::kMissileExt) << 24,
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    };
    if constexpr (SPLIT) {  // the tile's missile wave did all of that: wait for its word
      unsigned dw, spins = 0u;
      while (!((dw = (unsigned)__builtin_amdgcn_readfirstlane((int)hflags[1])) >> 31) && ++spins < kSpinLimit) __builtin_amdgcn_s_sleep(1);
      if (spins >= kSpinLimit && lane == 0u) atomicAdd(&a.acc[SF_ACC_HANDOVER], 1ull);
      asm volatile("" ::: "memory");
      wp = dw & 0x7FFFFFFFu;
    } else if (SF_ABL_PROJ == 0 || SF_ABL_PROJ == 3) {
#pragma unroll
      for (int r = 0; r < SF_MROWS; r++)
        if (m_live > 64u * r) m_row(prow[r].x, prow[r].y, pmeta[r], pcs[r], 64u * r + lane < m_live);
      if (m_live > 64u * SF_MROWS) {  // rare: more than SF_MROWS rows of live missiles in this tile
#pragma unroll 1
        for (unsigned r = SF_MROWS; 64u * r < m_live; r++) {
          // agent-scope loads: in a fused launch the previous tick of this wave wrote these rows
          const d2_t pr = __builtin_bit_cast(d2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, o.o16 + r * 1024u, SF_GOFF(missile_pos, 0), 16));
          const unsigned pm = __builtin_amdgcn_raw_buffer_load_b32(rs, o.o4 + r * 256u, SF_GOFF(missile_meta, 0), 16);
          m_row(pr.x, pr.y, pm, *reinterpret_cast<const d2_t*>(&trig[2 * SF_MM_ANGLE(pm)]), 64u * r + lane < m_live);
        }
      }
      if (__ballot(new_m_slot >= 0) != 0ull)
        m_row(new_m_x, new_m_y, SF_MM_PACK(new_m_angle, lane, new_m_slot & 31),
              *reinterpret_cast<const d2_t*>(&trig[2 * new_m_angle]), new_m_slot >= 0);
    }
    L.mpool = wp;
    // the owners collect what happened to their missiles (LDS is in order per wave; the fences pin the compiler)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long evp = evw[lane];
    if (FUSED) evw[lane] = 0ull;  // ready for the next tick
    unsigned ev_hit = (unsigned)evp, ev_out = (unsigned)(evp >> 32);
    dr_proj |= (ev_hit >> 24) & 0xFu;  // (draw records: this env's missiles near the score / the bar, sfd::hud_flags_near)
    ev_hit &= L.mmask;             // hit = live & collided
    ev_out &= L.mmask & ~ev_hit;   // out = live & !hit & outside
    unsigned ev = ev_hit | ev_out;
    hit_count = __popc(ev_hit);
    L.mmask &= ~ev;
    // slot order (SRC/game.cpp:355): one event per lane and round, rounds while any lane has one left; the
    // fortress state machine of :357-399 as selects (a lane without an event in this round has bit == 0)
    while (__ballot(ev != 0u) != 0ull) {
      const unsigned bit = ev & (0u - ev);
      ev &= ~bit;
      const bool on = bit != 0u, h = (ev_hit & bit) != 0u;
      const bool hv = h && (L.fl & SF_FL_FORT_ALIVE);        // a hit on a live fortress (:358)
      const int inc = hv && L.fort_vuln_t >= sfc::vuln_time;  // :359-363
      const int low = hv && !inc;                             // inside the vulnerability window (:364-384)
      const int destroy = low && L.vlner >= sfc::vuln_threshold + 1;
      L.vlner = inc ? L.vlner + 1 : (low ? 0 : L.vlner);
      S.vlner_incs += inc;
      S.max_vlner = (inc && L.vlner > S.max_vlner) ? L.vlner : S.max_vlner;
      L.fl &= destroy ? ~SF_FL_FORT_ALIVE : ~0u;
      L.fort_death_t = destroy ? 0 : L.fort_death_t;
      score(destroy ? sfc::Score<SHAPED>::destroy_reward : 0.0f, rew, L);
      S.destroyed += destroy;
      S.resets += low && !destroy;
      L.fort_vuln_t = hv ? 0 : L.fort_vuln_t;
      const int miss = on && !h;                              // left the game area (:394-398); miss_penalty is 0
      score(miss ? -sfc::Score<SHAPED>::miss_penalty : 0.0f, rew, L);
      S.missed += miss;
    }
  }
  SF_STAMP(6, false);

  // ---- stepTimers (SRC/game.cpp:425-451)
  L.fort_t += sfc::tick_ms;
  L.fort_death_t += sfc::tick_ms;
  L.fort_vuln_t += sfc::tick_ms;
  L.death_t += sfc::tick_ms;
  // (a key edge of processKeyState zeroes its timer, :227,231,235,244,250,253,257,261: applied here, nothing in between reads them)
  L.fire_t = ((key_edge_bits & 1u) ? 0 : L.fire_t) + ((L.fl & SF_FL_FIRE) ? 1 : -1);
  L.thrust_t = ((key_edge_bits & 2u) ? 0 : L.thrust_t) + ((L.fl & SF_FL_THRUST) ? 1 : -1);
  L.left_t = ((key_edge_bits & 4u) ? 0 : L.left_t) + ((L.fl & SF_FL_LEFT) ? 1 : -1);
  L.right_t = ((key_edge_bits & 8u) ? 0 : L.right_t) + ((L.fl & SF_FL_RIGHT) ? 1 : -1);

  int r = (int)rew;  // `return mReward` through `int stepOneTick` (SRC/game.hh:138): truncation
  // kept for `Game.step_one_tick`'s return value (field last_reward): the spare byte next to the flags
  L.fl = (L.fl & 0xFFu) | ((unsigned)(r & 0xFF) << 8);

  // ================= SSF_Env.step epilogue (ENV:233-253) =================
  const int fort_kill = r > 0;
  if (SHAPED) {
    const int vlner_change = L.vlner - L.prev_vlner;
    if (L.vlner <= 10 && !fort_kill) r += vlner_change;
    r = r > 1 ? 1 : (r < -1 ? -1 : r);
    r = r + 2 * fort_kill;
    L.prev_vlner = L.vlner;
  }
  L.time += sfc::tick_ms;                     // updateTime (SRC/game.cpp:475); nothing in between reads it
  const int done = L.time >= sfc::game_time;  // Game::isGameOver (SRC/game.cpp:487-489)

  // ---- optional telemetry: what happened this tick, as a bitmask (the reference's addEvent strings,
  //      SRC/game.cpp:124-127 and its call sites; include/sfmi.h SF_EV_*).  Everything it needs is already live.
  unsigned evmask = 0;
  if (a.events) {
    const int live_hits = S.vlner_incs + S.destroyed + S.resets;
    evmask = key_edges | (new_m_slot >= 0 ? SF_EV_MISSILE_FIRED : 0u) | (will_respawn ? SF_EV_SHIP_RESPAWN : 0u) |
             (S.big_hex_deaths ? SF_EV_EXPLODE_BIGHEX : 0u) | (S.small_hex_deaths ? SF_EV_EXPLODE_SMALLHEX : 0u) |
             (fort_respawned ? SF_EV_FORTRESS_RESPAWN : 0u) | (new_s_slot >= 0 ? SF_EV_FORTRESS_FIRED : 0u) |
             (S.shell_deaths ? SF_EV_SHELL_HIT_SHIP : 0u) | (live_hits ? SF_EV_HIT_FORTRESS : 0u) |
             (S.vlner_incs ? SF_EV_VLNER_INCREASED : 0u) | (S.destroyed ? SF_EV_FORTRESS_DESTROYED : 0u) |
             (S.resets ? SF_EV_VLNER_RESET : 0u) | ((int)hit_count > live_hits ? SF_EV_HIT_DEAD_FORTRESS : 0u) |
             (S.missed ? SF_EV_MISSILE_LEFT : 0u) | (done ? SF_EV_GAME_OVER : 0u);
  }

  SF_STAMP(10, false);
  // ================= statistics and the vec-env worker's auto-reset (rl/train.py:80-88) ======
  // The tick's share of the per-episode counters (they ride above the timers, vlner, time and the cursor: sf_layout.h)
  L.ep_return += r;
  L.ep_kills += (unsigned)fort_kill;
  L.c_resets += (unsigned)S.resets;
  L.c_missed += (unsigned)S.missed;
  L.c_incs += (unsigned)S.vlner_incs;
  L.c_maxv = (unsigned)S.max_vlner > L.c_maxv ? (unsigned)S.max_vlner : L.c_maxv;  // counter 12: a running maximum
  L.c_big += (unsigned)S.big_hex_deaths;
  L.c_small += (unsigned)S.small_hex_deaths;
  L.c_shell += (unsigned)S.shell_deaths;
  L.c_destroyed += (unsigned)S.destroyed;
  if (XTRA && !a.auto_reset) {  // uniform
    // The packed per-episode counters (sf_layout.h: SF_W_*) are sized for ONE episode.  A batch without auto-reset
    // keeps ticking past game over like the bare SSF_Env / Game (ENV:246) until the caller resets, and the reference's
    // plain ints keep counting: a field that no longer fits its bits is counted here (sticky: sf_check_state) instead
    // of wrapping silently.  Auto-resetting batches zero every field at 5 295 ticks, long before any of them fills up.
    const unsigned ov = ((L.c_resets | L.c_missed) >> 16) | ((L.c_incs | L.c_maxv | (unsigned)L.vlner) >> 12) |
                        ((L.c_big | L.c_small | L.c_shell | L.c_destroyed | L.ep_kills) >> 8) | ((unsigned)L.time >> 24) |
                        (((unsigned)(L.fire_t + 32768) | (unsigned)(L.thrust_t + 32768) | (unsigned)(L.left_t + 32768) |
                          (unsigned)(L.right_t + 32768)) >> 16) |
                        (unsigned)(S.shots && (L.kc0 & 0xFFFFu) == 0u) | (unsigned)(S.thrusts && (L.kc0 >> 16) == 0u) |
                        (unsigned)(S.lefts && (L.kc1 & 0xFFFFu) == 0u) | (unsigned)(S.rights && (L.kc1 >> 16) == 0u);
    if (__ballot(ov != 0u) != 0ull)
      if (ov != 0u && real) atomicAdd(&a.acc[SF_ACC_OVERFLOW], 1ull);
  }
  if (done && a.auto_reset) {
    // episode totals (rl/train.py:81-88,161-164), this tick's share included
    const int ep_ret = L.ep_return, ep_kil = (int)L.ep_kills;
    const int deaths = (int)(L.c_big + L.c_small + L.c_shell);  // killShip's three call sites
    const int shots = (int)(L.kc0 & 0xFFFFu);  // this tick's press included
    if (real) {
      atomicAdd(&a.acc[0], 1ull);
      atomicAdd(&a.acc[1], (unsigned long long)(long long)ep_ret);
      atomicAdd(&a.acc[2], (unsigned long long)((long long)ep_ret * ep_ret));
      atomicAdd(&a.acc[3], (unsigned long long)(long long)ep_kil);
      atomicAdd(&a.acc[4], (unsigned long long)deaths);
      atomicAdd(&a.acc[5], (unsigned long long)shots);
      atomicMin((long long*)&a.acc[6], (long long)ep_ret);
      atomicMax((long long*)&a.acc[7], (long long)ep_ret);
    }
    // a new Game (ENV:163-178).  Its missiles are gone with it: the pool entries this lane still owns leave below
    new_game(a, L);
    a_pos = sf_atan2<true>(L.sy - sfc::fort_y, L.sx - sfc::fort_x);
    a_vel = sf_atan2<false>(L.vy, L.vx);
  }
  // A lane that started a new game while others of its tile play on (only possible when episodes are out of step:
  // sf_set_field, a reset of part of a tile) must take its entries out of the shared pool.  With every lane of the
  // tile done in the same tick -- the regular case -- the pool is simply emptied.
  {
    const unsigned long long dn = __ballot(done && a.auto_reset);
    if (dn != 0ull) {
      if (dn == ~0ull) {
        L.mpool = 0;
      } else {
        const unsigned n_before = (unsigned)__builtin_amdgcn_readfirstlane((int)L.mpool);
        unsigned wp = 0;
        if constexpr (SPLIT) {  // the missile wave's stores of this tick's rows are acknowledged
          unsigned spins = 0u;
          while (__builtin_amdgcn_readfirstlane((int)hflags[2]) == 0 && ++spins < kSpinLimit) __builtin_amdgcn_s_sleep(1);
          if (spins >= kSpinLimit && lane == 0u) atomicAdd(&a.acc[SF_ACC_HANDOVER], 1ull);
          asm volatile("" ::: "memory");
        }
#pragma unroll 1
        for (unsigned r = 0; 64u * r < n_before; r++) {
          const u4_t pr = __builtin_amdgcn_raw_buffer_load_b128(rs, o.o16 + r * 1024u, SF_GOFF(missile_pos, 0), 16);
          const unsigned pm = __builtin_amdgcn_raw_buffer_load_b32(rs, o.o4 + r * 256u, SF_GOFF(missile_meta, 0), 16);
          const bool keep = (64u * r + lane < n_before) && !((dn >> SF_MM_OWNER(pm)) & 1ull);
          const unsigned long long kb = __ballot(keep);
          const unsigned idx = wp + __builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0u));
          wp += (unsigned)__popcll(kb);
          sf_buf_st128<kStAux>(pr, rs, keep ? idx * 16u : SF_OOB, SF_GOFF(missile_pos, 0));
          __builtin_amdgcn_raw_buffer_store_b32(pm, rs, keep ? idx * 4u : SF_OOB, SF_GOFF(missile_meta, 0), kStAux);
        }
        L.mpool = wp;
      }
    }
  }

  SF_STAMP(11, false);
  if (draw_now) {  // uniform: the env's draw record (sf_drawrec.h) -- header, ship, fortress; the missiles' entries went out above
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const sfd::Header h = sfd::make_header(L.sx, L.sy, L.angle, (L.fl & SF_FL_SHIP_ALIVE) != 0u, (L.fl & SF_FL_FORT_ALIVE) != 0u, L.fort_angle,
                                           L.points, L.vlner, L.fort_vuln_t, L.mmask, L.smask,
                                           (done && a.auto_reset) ? 0u : dr_proj /* a new game has no projectiles */, a.draw_pics != 0, L.time);
    const __amdgpu_buffer_rsrc_t rs_dr = __builtin_amdgcn_make_buffer_rsrc(
        a.draw + (size_t)__builtin_amdgcn_readfirstlane(i >> 6) * (size_t)SF_DR_TILE_BYTES, 0, SF_DR_TILE_BYTES, 0x00020000);
    const unsigned d0 = real ? lane * (unsigned)SF_DR_LANE_STRIDE : SF_OOB;
    sf_buf_st128<SF_DR_AUX>(u4_t{h.w[0], h.w[1], h.w[2], h.w[3]}, rs_dr, d0, 0);
    sf_buf_st128<SF_DR_AUX>(u4_t{h.w[4], h.w[5], h.w[6], h.w[7]}, rs_dr, d0, SF_DR_PIECE_STRIDE);
    sf_buf_st128<SF_DR_AUX>(__builtin_bit_cast(u4_t, (d2_t{L.sx, L.sy})), rs_dr, d0, (SF_DR_PIECE_OBJ0 + SF_DR_OBJ_SHIP) * SF_DR_PIECE_STRIDE);
  }
  if (!FUSED) store_lane_buf(rs, o, L);
  SF_STAMP(14, false);

  if (real) {
    if (reward_out) SF_ST(int32_t, (unsigned char*)(reward_out + so), g.o4, r);
    if (done_out) SF_ST(uint8_t, (unsigned char*)(done_out + so), g.o1, (uint8_t)done);
    if (info_out) SF_ST(uint8_t, (unsigned char*)(info_out + so), g.o1, (uint8_t)fort_kill);
    if (a.events) SF_ST(uint32_t, (unsigned char*)(a.events + so), g.o4, evmask);
    if (XTRA && a.act_out) SF_ST(uint8_t, (unsigned char*)(a.act_out + so), g.o1, (uint8_t)act_raw);  // what the lane played (sampled or given)
    SF_STAMP(15, false);
    if (a.t_reward) {  // uniform; the same float32 operations in the same order as rl/train.py:82-88
      const float rf = (float)r, mask = done ? 0.0f : 1.0f;
      SF_ST(float, (unsigned char*)(a.t_reward + so), g.o4, rf);
      if (a.t_mask) SF_ST(float, (unsigned char*)(a.t_mask + so), g.o4, mask);
      if (a.t_episode) {
        const float ep = SF_LD(float, (const unsigned char*)a.t_episode, g.o4) + rf;  // episode_rewards += reward
        if (a.t_final) {
          float fin = SF_LD(float, (const unsigned char*)a.t_final, g.o4) * mask;     // final_rewards *= masks
          fin = fin + (1.0f - mask) * ep;                                              // += (1 - masks) * episode_rewards
          SF_ST(float, (unsigned char*)a.t_final, g.o4, fin);
        }
        SF_ST(float, (unsigned char*)a.t_episode, g.o4, ep * mask);                    // episode_rewards *= masks
      }
      if (a.t_actions) SF_ST(long long, (unsigned char*)(a.t_actions + so), g.o8, (long long)act_raw);
    }
  }
  SF_STAMP(7, false);
  if constexpr (SF_ABL_OBS == 1) {  // timing-only: no observation at all
  } else if constexpr (OBSK == 1) {  // the host guarantees: features, float32, n_envs % 64 == 0, aligned output (sf_launch_step)
    constexpr int DIM = AUTOTURN ? 17 : 19;
    float* stage = reinterpret_cast<float*>(lds + kLdsStage);
    if (SF_ABL_OBS != 3) {  // SF_ABL_OBS 3 (timing-only): the stores alone, of whatever the staging rows hold
      const Extras e = compute_extras(a, L, a_pos, a_vel);
      write_features_f32<DIM>(stage + tid * DIM, L, e, a.real_shell_count);
    }
    if ((i & ~63u) < (unsigned)n_envs_p)  // the padding waves behind the batch write nothing
      flush_features_f32<DIM>(stage + (tid & ~63u) * DIM, (float*)obs + (so + (i & ~63u)) * DIM, lane);
  } else if (obs != nullptr && a.obs_type != 3) {  // uniform across the grid
    Extras e = compute_extras(a, L, a_pos, a_vel);
    if (a.ref_reset_obs && done && a.auto_reset) e = Extras{0.0, 0.0, 0.0};  // (a new game's observation as the reference returns it: sf_launch_step keeps such batches here)
    if (a.obs_f64) {
      double* stage = lds + kLdsStage;
      write_obs<double>(a, stage + tid * a.obs_dim, L, e);
      flush_obs_wave<double>(a, stage + (tid & ~63u) * a.obs_dim, (double*)obs + so * a.obs_dim, i & ~63u, lane,
                             obs_vec_ok);
    } else {
      float* stage = reinterpret_cast<float*>(lds + kLdsStage);
      write_obs<float>(a, stage + tid * a.obs_dim, L, e);
      flush_obs_wave<float>(a, stage + (tid & ~63u) * a.obs_dim, (float*)obs + so * a.obs_dim, i & ~63u, lane,
                            obs_vec_ok);
    }
  }
  if constexpr (OBSK != 1) {
    // image batches: which ships died this tick, one word per tile.  sf_render_kernel starts those frames first (the
    // first frame of an explosion costs three ordinary ones, sf_render.hip); a scheduling hint, nothing else reads it
    if (a.hint) {  // uniform
      const unsigned long long died = __ballot(real && (S.big_hex_deaths | S.small_hex_deaths | S.shell_deaths) != 0);
      if (lane == 0) a.hint[i >> 6] = died;
    }
  }
  if (!FUSED && a.n_partials) {  // uniform: VecNormalize's reduction rides on the step (sf_step_normalize)
    double my_ret = 0;
    if (a.n_ret && real) {
      my_ret = SF_LD(double, (const unsigned char*)a.n_ret, g.o8) * a.n_gamma + (double)r;  // ret = ret * gamma + rews
      SF_ST(double, (unsigned char*)a.n_ret, g.o8, my_ret);
    }
    const bool has_obs = obs != nullptr && a.obs_type != 3;
    if (a.obs_f64)
      norm_partials_wave<double>(a, lds + kLdsStage + (size_t)(tid & ~63u) * a.obs_dim, i & ~63u, lane, my_ret, has_obs);
    else
      norm_partials_wave<float>(a, reinterpret_cast<float*>(lds + kLdsStage) + (size_t)(tid & ~63u) * a.obs_dim, i & ~63u,
                                lane, my_ret, has_obs);
  }
  }  // tick loop
  if (FUSED) store_lane_buf(rs, o, L);
  if (XTRA && act_type == SF_ACT_SAMPLED && lane == 0)  // the tile's tick counter moves on by the ticks of this launch
    reinterpret_cast<unsigned*>(const_cast<void*>(actions))[4 * (size_t)(i >> 6)] = act_rec.x + (unsigned)n_iter;
  SF_STAMP(8, false);
  SF_STAMP(9, true);
#ifdef SF_STAMPS
  stamp_[13] = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (a.dbg != nullptr && (tid & 63) == 0) {
    unsigned long long* d = a.dbg + (size_t)(i >> 6) * 16;
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = stamp_[k];
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// The envs' draw records (sf_drawrec.h) from the state as it is in HBM: what the image instantiations of the step kernel
// leave behind themselves, for a state that got there any other way -- a reset, sf_set_field, a batch that steps with a
// symbolic observation and renders now and then.  One wave per tile, a lane per env; the pool's entries file their
// transforms at [owner][slot] and tell their owners through LDS where they come near the score / the bar, exactly like the
// step kernel's m_row.  Same functions, same values: tests/test_gpu_image.py compares the two byte for byte.
__global__ __launch_bounds__(64) void sf_drawrec_kernel(const unsigned char* state, const double* consts, int n_envs,
                                                       unsigned char* draw, int pics) {
  __shared__ unsigned near[64];
  typedef float f4_t __attribute__((ext_vector_type(4)));
  const unsigned lane = threadIdx.x;
  const long tile_i = blockIdx.x;
  const unsigned char* tb = state + tile_i * sfl::kTileBytes;
  unsigned char* const dr = draw + tile_i * (long)SF_DR_TILE_BYTES;
  const unsigned o16 = lane * 16u;
  const d2_t sp = SF_LD(d2_t, SF_CHUNK(ship_pos, 0), o16);
  const i4_t tc = SF_LD(i4_t, SF_CHUNK(timers_b, 0), o16);
  const i4_t sc = SF_LD(i4_t, SF_CHUNK(score, 0), o16);
  const i4_t mi = SF_LD(i4_t, SF_CHUNK(misc, 0), o16);
  const i4_t sm = SF_LD(i4_t, SF_CHUNK(small, 0), o16);
  near[lane] = 0u;
  __syncthreads();
  const unsigned n_pool = (unsigned)__builtin_amdgcn_readfirstlane(mi.z) >> SF_MPOOL_SHIFT;  // (the same in every lane of the tile)
  for (unsigned k = lane; k < n_pool; k += 64) {
    const d2_t p = SF_LD(d2_t, SF_CHUNK(missile_pos, 0), k * 16u);
    const unsigned m = SF_LD(unsigned, SF_CHUNK(missile_meta, 0), k * 4u);
    *reinterpret_cast<d2_t*>(dr + (SF_DR_PIECE_OBJ0 + SF_DR_OBJ_MISSILE0 + SF_MM_SLOT(m)) * SF_DR_PIECE_STRIDE + SF_MM_OWNER(m) * SF_DR_LANE_STRIDE) = p;
    *reinterpret_cast<int16_t*>(dr + SF_DR_ANGLES_OFF + 2 * SF_MM_SLOT(m) + SF_MM_OWNER(m) * SF_DR_LANE_STRIDE) = (int16_t)SF_MM_ANGLE(m);
    const unsigned f = sfd::hud_flags_near((float)p.x, (float)p.y, sfd::kMissileExt);
    if (f) atomicOr(&near[SF_MM_OWNER(m)], f);
  }
  const unsigned smask = (unsigned)mi.w & SF_MASK_LOW, mmask = (unsigned)mi.z & SF_MASK_LOW;
  unsigned proj = 0u;
  for (unsigned rest = smask; rest; rest &= rest - 1u) {
    const int s = __ffs(rest) - 1;
    const d2_t q = SF_LD(d2_t, SF_CHUNK(shell_pos, s), o16);
    proj |= sfd::hud_flags_near((float)q.x, (float)q.y, sfd::kShellExt);
  }
  __syncthreads();
  proj |= near[lane];
  if (tile_i * 64 + lane >= n_envs) return;
  const int ship_angle = (int16_t)(sm.x & 0xFFFF), fort_angle = (int16_t)((unsigned)sm.x >> 16);
  const unsigned fl = ((unsigned)sm.y >> 16) & 0xFFu;
  const sfd::Header h = sfd::make_header(sp.x, sp.y, ship_angle, (fl & SF_FL_SHIP_ALIVE) != 0u, (fl & SF_FL_FORT_ALIVE) != 0u, fort_angle,
                                         __int_as_float(sc.x), sc.z & 0xFFF, tc.w, mmask, smask, proj, pics != 0,
                                         (int)((unsigned)sc.w & 0xFFFFFFu));
  unsigned char* const me = dr + lane * SF_DR_LANE_STRIDE;
  *reinterpret_cast<u4_t*>(me) = u4_t{h.w[0], h.w[1], h.w[2], h.w[3]};
  *reinterpret_cast<u4_t*>(me + SF_DR_PIECE_STRIDE) = u4_t{h.w[4], h.w[5], h.w[6], h.w[7]};
  *reinterpret_cast<d2_t*>(me + (SF_DR_PIECE_OBJ0 + SF_DR_OBJ_SHIP) * SF_DR_PIECE_STRIDE) = sp;
  (void)consts;
}

hipError_t sf_launch_drawrec(const SfKernelArgs& a, hipStream_t stream) {
  if (!a.draw) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sf_drawrec_kernel, dim3((unsigned)(a.lanes / 64)), dim3(64), 0, stream, a.state, a.consts, a.n_envs, a.draw,
                     a.draw_pics);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// sf_get_field / sf_set_field: one field between the tiled state and a linear [count][n_envs]
// buffer (not on the hot path).
template <typename T>
__global__ __launch_bounds__(SF_BLOCK) void sf_field_copy_kernel(unsigned char* state, int n_envs, long tile_off,
                                                                int lane_stride, int slot_stride, int count,
                                                                T* linear, int to_linear) {
  const long e = (long)blockIdx.x * SF_BLOCK + threadIdx.x;
  if (e >= n_envs) return;
  unsigned char* lane0 = state + (e >> 6) * sfl::kTileBytes + tile_off + (e & 63) * lane_stride;
  for (int c = 0; c < count; c++) {
    T* p = reinterpret_cast<T*>(lane0 + (long)c * slot_stride);
    if (to_linear)
      linear[(long)c * n_envs + e] = *p;
    else
      *p = linear[(long)c * n_envs + e];
  }
}

// sf_get_field / sf_set_field for the fields that are not one element at a fixed place of a chunk (sf_layout.h: SF_FK_*).

// "stats": the reference's 13 ints (SRC/game.hh:29-43) from / to their bit fields (sf_layout.h: SF_W_*); ship deaths
// (row 3) is the sum of rows 0-2 and is not stored (a value written to it is ignored).
__global__ __launch_bounds__(SF_BLOCK) void sf_stats_copy_kernel(unsigned char* state, int n_envs, int32_t* linear,
                                                                int to_linear) {
  const long e = (long)blockIdx.x * SF_BLOCK + threadIdx.x;
  if (e >= n_envs) return;
  unsigned char* tile = state + (e >> 6) * sfl::kTileBytes;
  const long lo = (e & 63) * 16;
  uint16_t* kc = reinterpret_cast<uint16_t*>(tile + sfl::chunk_offset(SF_G_small, 0) + lo + SF_KEYCOUNT_BYTE);
  uint32_t* ta = reinterpret_cast<uint32_t*>(tile + sfl::chunk_offset(SF_G_timers_a, 0) + lo);  // pvl, fire, thrust, left
  uint32_t* sc = reinterpret_cast<uint32_t*>(tile + sfl::chunk_offset(SF_G_score, 0) + lo);     // .., .., vlner, time
  uint32_t* mi = reinterpret_cast<uint32_t*>(tile + sfl::chunk_offset(SF_G_misc, 0) + lo);      // .., cursor, .., ..
#define SF_ROW(k) linear[(long)(k) * n_envs + e]
#define SF_PUT(word, shift, bits, k) word = (word & ~((((1u << (bits)) - 1u)) << (shift))) | (((uint32_t)SF_ROW(k) & ((1u << (bits)) - 1u)) << (shift))
  if (to_linear) {
    const int big = (int)(ta[0] >> 24), sml = (int)(sc[2] >> 24), shl = (int)(sc[3] >> 24);
    SF_ROW(SF_ST_BIG_HEX_DEATHS) = big;
    SF_ROW(SF_ST_SMALL_HEX_DEATHS) = sml;
    SF_ROW(SF_ST_SHELL_DEATHS) = shl;
    SF_ROW(SF_ST_SHIP_DEATHS) = big + sml + shl;
    SF_ROW(SF_ST_RESETS) = (int)(ta[1] >> 16);
    SF_ROW(SF_ST_DESTROYED) = (int)(mi[1] >> 24);
    SF_ROW(SF_ST_MISSED) = (int)(ta[2] >> 16);
    for (int c = 0; c < SF_ST_KEY_COUNT; c++) SF_ROW(SF_ST_KEY_FIRST + c) = kc[c];
    SF_ROW(SF_ST_VLNER_INCS) = (int)((ta[0] >> 12) & 0xFFFu);
    SF_ROW(SF_ST_MAX_VLNER) = (int)((sc[2] >> 12) & 0xFFFu);
  } else {
    SF_PUT(ta[0], 24, 8, SF_ST_BIG_HEX_DEATHS);
    SF_PUT(sc[2], 24, 8, SF_ST_SMALL_HEX_DEATHS);
    SF_PUT(sc[3], 24, 8, SF_ST_SHELL_DEATHS);
    SF_PUT(ta[1], 16, 16, SF_ST_RESETS);
    SF_PUT(mi[1], 24, 8, SF_ST_DESTROYED);
    SF_PUT(ta[2], 16, 16, SF_ST_MISSED);
    for (int c = 0; c < SF_ST_KEY_COUNT; c++) kc[c] = (uint16_t)SF_ROW(SF_ST_KEY_FIRST + c);
    SF_PUT(ta[0], 12, 12, SF_ST_VLNER_INCS);
    SF_PUT(sc[2], 12, 12, SF_ST_MAX_VLNER);
  }
#undef SF_PUT
#undef SF_ROW
}

// a bit field of a 32-bit word of the lane's chunk (sf_layout.h: SF_BITFIELDS) from / to a linear int32 buffer
__global__ __launch_bounds__(SF_BLOCK) void sf_bits_copy_kernel(unsigned char* state, int n_envs, long tile_off, int shift,
                                                               int bits, int is_signed, uint32_t* linear, int to_linear) {
  const long e = (long)blockIdx.x * SF_BLOCK + threadIdx.x;
  if (e >= n_envs) return;
  uint32_t* w = reinterpret_cast<uint32_t*>(state + (e >> 6) * sfl::kTileBytes + tile_off + (e & 63) * 16);
  const uint32_t mask = bits >= 32 ? ~0u : ((1u << bits) - 1u);
  if (to_linear) {
    uint32_t v = (*w >> shift) & mask;
    if (is_signed && bits < 32 && (v >> (bits - 1))) v |= ~mask;
    linear[e] = v;
  } else {
    *w = (*w & ~(mask << shift)) | ((linear[e] & mask) << shift);
  }
}
// "ep_return": int32, bits 0..15 above the left timer, bits 16..31 above the right timer
__global__ __launch_bounds__(SF_BLOCK) void sf_epret_copy_kernel(unsigned char* state, int n_envs, int32_t* linear, int to_linear) {
  const long e = (long)blockIdx.x * SF_BLOCK + threadIdx.x;
  if (e >= n_envs) return;
  unsigned char* tile = state + (e >> 6) * sfl::kTileBytes;
  uint32_t* wl = reinterpret_cast<uint32_t*>(tile + sfl::chunk_offset(SF_G_timers_a, 0) + (e & 63) * 16 + 12);
  uint32_t* wr = reinterpret_cast<uint32_t*>(tile + sfl::chunk_offset(SF_G_timers_b, 0) + (e & 63) * 16);
  if (to_linear) {
    linear[e] = (int32_t)((*wl >> 16) | (*wr & 0xFFFF0000u));
  } else {
    const uint32_t v = (uint32_t)linear[e];
    *wl = (*wl & 0xFFFFu) | (v << 16);
    *wr = (*wr & 0xFFFFu) | (v & 0xFFFF0000u);
  }
}

// The missile fields, per env and slot as the reference has them (mMissiles[i], SRC/game.hh:90), from / to the tile's pool.
// `slots` is the batch's slot-major view [SF_NSLOT][n_envs] of (x, y) as d2_t and of the heading as int32.
// Pool -> slots: every live entry goes to (slot, owner); slots without a missile read 0.
__global__ __launch_bounds__(64) void sf_mpool_to_slots_kernel(const unsigned char* state, int n_envs, d2_t* sl_pos,
                                                              int32_t* sl_ang) {
  const long tile_i = blockIdx.x;
  const unsigned lane = threadIdx.x;
  const unsigned char* tile = state + tile_i * sfl::kTileBytes;
  const long e = tile_i * 64 + lane;
  if (e < n_envs)
    for (int s = 0; s < SF_NSLOT; s++) {
      sl_pos[(long)s * n_envs + e] = d2_t{0, 0};
      sl_ang[(long)s * n_envs + e] = 0;
    }
  __syncthreads();
  const unsigned n = *reinterpret_cast<const uint32_t*>(tile + sfl::chunk_offset(SF_G_misc, 0) + 8) >> SF_MPOOL_SHIFT;
  for (unsigned k = lane; k < n; k += 64) {
    const d2_t p = *reinterpret_cast<const d2_t*>(tile + sfl::chunk_offset(SF_G_missile_pos, 0) + (size_t)k * 16);
    const unsigned m = *reinterpret_cast<const uint32_t*>(tile + sfl::chunk_offset(SF_G_missile_meta, 0) + (size_t)k * 4);
    const long oe = tile_i * 64 + SF_MM_OWNER(m);
    if (oe < n_envs) {
      sl_pos[(long)SF_MM_SLOT(m) * n_envs + oe] = p;
      sl_ang[(long)SF_MM_SLOT(m) * n_envs + oe] = (int)SF_MM_ANGLE(m);
    }
  }
}
// Slots -> pool: the tile's pool is rebuilt from the alive masks, slot by slot (ballot + prefix count, the step
// kernel's compaction), and the count written into every lane's missile word.
__global__ __launch_bounds__(64) void sf_slots_to_mpool_kernel(unsigned char* state, int n_envs, const d2_t* sl_pos,
                                                              const int32_t* sl_ang) {
  const long tile_i = blockIdx.x;
  const unsigned lane = threadIdx.x;
  unsigned char* tile = state + tile_i * sfl::kTileBytes;
  const long e = tile_i * 64 + lane;
  uint32_t* mw = reinterpret_cast<uint32_t*>(tile + sfl::chunk_offset(SF_G_misc, 0) + lane * 16 + 8);
  const unsigned mask = e < n_envs ? (*mw & SF_MASK_LOW) : 0u;
  unsigned wp = 0;
  for (int s = 0; s < SF_NSLOT; s++) {
    const bool live = (mask >> s) & 1u;
    const unsigned long long b = __ballot(live);
    const unsigned idx = wp + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    wp += (unsigned)__popcll(b);
    if (live) {
      *reinterpret_cast<d2_t*>(tile + sfl::chunk_offset(SF_G_missile_pos, 0) + (size_t)idx * 16) = sl_pos[(long)s * n_envs + e];
      *reinterpret_cast<uint32_t*>(tile + sfl::chunk_offset(SF_G_missile_meta, 0) + (size_t)idx * 4) =
          SF_MM_PACK((unsigned)sl_ang[(long)s * n_envs + e] & 511u, lane, s);
    }
  }
  *mw = mask | (wp << SF_MPOOL_SHIFT);
}

hipError_t sf_launch_mpool_to_slots(const unsigned char* state, int n_envs, void* sl_pos, int32_t* sl_ang, hipStream_t stream) {
  hipLaunchKernelGGL(sf_mpool_to_slots_kernel, dim3((unsigned)((n_envs + 63) / 64)), dim3(64), 0, stream, state, n_envs,
                     (d2_t*)sl_pos, sl_ang);
  return hipGetLastError();
}
hipError_t sf_launch_slots_to_mpool(unsigned char* state, long lanes, int n_envs, const void* sl_pos, const int32_t* sl_ang,
                                    hipStream_t stream) {
  hipLaunchKernelGGL(sf_slots_to_mpool_kernel, dim3((unsigned)(lanes / 64)), dim3(64), 0, stream, state, n_envs,
                     (const d2_t*)sl_pos, sl_ang);
  return hipGetLastError();
}

// one component of the slot-major missile view <-> the caller's linear [SF_NSLOT][n_envs] buffer
// which: 0 = x, 1 = y (double), 2 = heading (int16)
__global__ __launch_bounds__(SF_BLOCK) void sf_mslot_component_kernel(d2_t* sl_pos, int32_t* sl_ang, long total, int which,
                                                                     void* linear, int to_linear) {
  const long k = (long)blockIdx.x * SF_BLOCK + threadIdx.x;
  if (k >= total) return;
  if (which == 2) {
    if (to_linear) ((int16_t*)linear)[k] = (int16_t)sl_ang[k];
    else sl_ang[k] = ((const int16_t*)linear)[k];
  } else {
    double* comp = reinterpret_cast<double*>(sl_pos + k) + which;
    if (to_linear) ((double*)linear)[k] = *comp;
    else *comp = ((const double*)linear)[k];
  }
}
hipError_t sf_launch_mslot_component(void* sl_pos, int32_t* sl_ang, long total, int which, void* linear, int to_linear,
                                     hipStream_t stream) {
  hipLaunchKernelGGL(sf_mslot_component_kernel, dim3((unsigned)((total + SF_BLOCK - 1) / SF_BLOCK)), dim3(SF_BLOCK), 0, stream,
                     (d2_t*)sl_pos, sl_ang, total, which, linear, to_linear);
  return hipGetLastError();
}

// PMC calibration (sf_calibration_copy): copy whole 16-byte chunks of one group to the linear
// buffer -- 16 B per lane, 1 KiB per wave-instruction, exactly the step kernel's access pattern.
__global__ __launch_bounds__(SF_BLOCK) void sf_group_copy_kernel(const unsigned char* state, int n_envs, long tile_off,
                                                                int slots, i4_t* linear) {
  const long e = (long)blockIdx.x * SF_BLOCK + threadIdx.x;
  if (e >= n_envs) return;
  const unsigned char* lane0 = state + (e >> 6) * sfl::kTileBytes + tile_off + (e & 63) * 16;
  for (int c = 0; c < slots; c++)
    linear[(long)c * n_envs + e] = *reinterpret_cast<const i4_t*>(lane0 + (long)c * 16 * sfl::kTileLanes);
}

hipError_t sf_launch_group_copy(const unsigned char* state, int n_envs, int group, unsigned char* linear,
                                hipStream_t stream) {
  const unsigned grid = (unsigned)((n_envs + SF_BLOCK - 1) / SF_BLOCK);
  hipLaunchKernelGGL(sf_group_copy_kernel, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs,
                     sfl::group_offset(group), sfl::kGroups[group].slots, (i4_t*)linear);
  return hipGetLastError();
}

hipError_t sf_launch_field_copy(unsigned char* state, int n_envs, int field, unsigned char* linear, int to_linear,
                                hipStream_t stream) {
  const sfl::FieldMeta& m = sfl::kFields[field];
  const unsigned grid = (unsigned)((n_envs + SF_BLOCK - 1) / SF_BLOCK);
  if (m.kind == SF_FK_STATS) {
    hipLaunchKernelGGL(sf_stats_copy_kernel, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs, (int32_t*)linear, to_linear);
    return hipGetLastError();
  }
  if (m.kind == SF_FK_BITS) {
    const sfl::BitField bf = sfl::bit_field(field);
    hipLaunchKernelGGL(sf_bits_copy_kernel, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs,
                       sfl::group_offset(m.group) + m.byte_in_chunk, bf.shift, bf.bits, bf.is_signed, (uint32_t*)linear, to_linear);
    return hipGetLastError();
  }
  if (m.kind == SF_FK_EPRET) {
    hipLaunchKernelGGL(sf_epret_copy_kernel, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs, (int32_t*)linear, to_linear);
    return hipGetLastError();
  }
  if (m.kind == SF_FK_MPOOL) return hipErrorInvalidValue;  // sf_capi.cpp goes through the slot view (sf_launch_mslot_component)
  const int lane_stride = sfl::kGroups[m.group].chunk, slot_stride = lane_stride * sfl::kTileLanes;
  const long off = sfl::group_offset(m.group) + m.byte_in_chunk;
  const int elem_size = m.elem_size, count = m.count;
  switch (elem_size) {
    case 1:
      hipLaunchKernelGGL(sf_field_copy_kernel<uint8_t>, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs, off,
                         lane_stride, slot_stride, count, (uint8_t*)linear, to_linear);
      break;
    case 2:
      hipLaunchKernelGGL(sf_field_copy_kernel<uint16_t>, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs, off,
                         lane_stride, slot_stride, count, (uint16_t*)linear, to_linear);
      break;
    case 4:
      hipLaunchKernelGGL(sf_field_copy_kernel<uint32_t>, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs, off,
                         lane_stride, slot_stride, count, (uint32_t*)linear, to_linear);
      break;
    default:
      hipLaunchKernelGGL(sf_field_copy_kernel<uint64_t>, dim3(grid), dim3(SF_BLOCK), 0, stream, state, n_envs, off,
                         lane_stride, slot_stride, count, (uint64_t*)linear, to_linear);
      break;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers (called by sf_capi.cpp)

hipError_t sf_launch_reset(const SfKernelArgs& a, int first, unsigned cursor0, unsigned stride, void* obs,
                           hipStream_t stream) {
  const unsigned grid = (unsigned)(a.lanes / SF_BLOCK);
  hipLaunchKernelGGL(sf_reset_kernel, dim3(grid), dim3(SF_BLOCK), 0, stream, a, first, cursor0, stride, obs);
  return hipGetLastError();
}

hipError_t sf_launch_step(const SfKernelArgs& a, bool autoturn, bool shaped, const void* actions, int act_type, void* obs,
                          int32_t* reward, uint8_t* done, uint8_t* info, int n_steps, bool fused, hipStream_t stream) {
  // threads per workgroup: the smallest of 64 / 128 / 256 that still makes a workgroup per CU (256 of them) -- see sf_step_kernel
  int blk = fused ? SF_BLOCK : (a.lanes <= 64 * 256 ? 64 : (a.lanes <= 128 * 256 ? 128 : SF_BLOCK));
  const size_t elem = a.obs_f64 ? sizeof(double) : sizeof(float);
  // 16-byte obs stores need every tick's row of the output to start 16-byte aligned
  const int vec_ok = ((uintptr_t)obs & 15u) == 0 && (!fused || ((size_t)a.n_envs * a.obs_dim * elem) % 16 == 0);
  // the default observation has its own instantiations (OBSK = 1): see write_features_f32
  const bool fast_obs = obs != nullptr && a.obs_type == 0 && !a.obs_f64 && vec_ok && a.n_envs % 64 == 0 &&
                        a.obs_dim == (autoturn ? 17 : 19) && !a.ref_reset_obs;
  const bool xtra = act_type == SF_ACT_SAMPLED || a.act_out != nullptr || !a.auto_reset;
  // a split launch (sf_step_kernel: a missile wave per tile) where a tile's wave is alone on its SIMD otherwise
  // (SF_SPLIT 2, for tests: every batch it can serve)
  // (SFMI_FORCE_SPLIT=1 in the environment, for tests: every batch the instantiation can serve; =2 says so once on stderr)
  static const int force_split = [] { const char* e = getenv("SFMI_FORCE_SPLIT"); return e ? atoi(e) : 0; }();
  const bool split = SF_SPLIT && !fused && fast_obs && !xtra && (a.lanes <= 65536 || SF_SPLIT == 2 || force_split) && a.draw == nullptr;
  if (split && force_split == 2) {
    static bool said = false;
    if (!said) fprintf(stderr, "sfmi: split launch (%ld lanes)\n", (long)a.lanes);
    said = true;
  }
  const unsigned grid = (unsigned)(a.lanes / blk);
  size_t lds_bytes = (size_t)(SF_LDS_DOUBLES + blk + SF_ATAB_DOUBLES) * sizeof(double) + (size_t)blk * a.obs_dim * elem;
  if (split) lds_bytes += (size_t)(2 * blk + blk / 2 + blk / 32) * sizeof(double);
#define SF_GO2(AT, SH, FU, OK, XT, BL)                                                                             \
  hipLaunchKernelGGL((sf_step_kernel<AT, SH, FU, OK, XT, BL>), dim3(grid), dim3((BL) > 1000 ? 2 * ((BL) - 1000) : (BL)), lds_bytes, stream, a.state,    \
                     a.consts, actions, a.n_envs, act_type, reward, done, info, a, obs, vec_ok, n_steps)
#define SF_GO1(AT, SH, FU, OK, XT)                                                                                 \
  do {                                                                                                             \
    if (FU || blk == SF_BLOCK) SF_GO2(AT, SH, FU, OK, XT, SF_BLOCK);                                               \
    else if (blk == 128) SF_GO2(AT, SH, false, OK, XT, 128);                                                       \
    else SF_GO2(AT, SH, false, OK, XT, 64);                                                                        \
  } while (0)
#if SF_SPLIT
#define SF_GO_SPLIT(AT, SH)                                                                                        \
  if (split) {                                                                                                     \
    if (blk == SF_BLOCK) SF_GO2(AT, SH, false, 1, false, 1256);                                                    \
    else if (blk == 128) SF_GO2(AT, SH, false, 1, false, 1128);                                                    \
    else SF_GO2(AT, SH, false, 1, false, 1064);                                                                    \
  } else
#else
#define SF_GO_SPLIT(AT, SH)
#endif
#define SF_GO(AT, SH, FU)                                                                                          \
  SF_GO_SPLIT(AT, SH)                                                                                              \
  if (fast_obs) {                                                                                                  \
    if (xtra) SF_GO1(AT, SH, FU, 1, true); else SF_GO1(AT, SH, FU, 1, false);                                      \
  } else {                                                                                                         \
    if (xtra) SF_GO1(AT, SH, FU, 0, true); else SF_GO1(AT, SH, FU, 0, false);                                      \
  }
  // the four presets of SRC/configs.cpp:51-89 are exactly (autoTurn) x (shaped scoring)
  const int sel = (autoturn ? 4 : 0) | (shaped ? 2 : 0) | (fused ? 1 : 0);
  switch (sel) {
    case 0: SF_GO(false, false, false); break;
    case 1: SF_GO(false, false, true); break;
    case 2: SF_GO(false, true, false); break;
    case 3: SF_GO(false, true, true); break;
    case 4: SF_GO(true, false, false); break;
    case 5: SF_GO(true, false, true); break;
    case 6: SF_GO(true, true, false); break;
    default: SF_GO(true, true, true); break;
  }
#undef SF_GO
#undef SF_GO1
#undef SF_GO_SPLIT
#undef SF_GO2
  return hipGetLastError();
}
