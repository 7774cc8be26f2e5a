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
//  * state is struct-of-arrays in HBM (sf_layout.h): a wave's 64 lanes read 64
//    consecutive elements of each field -- every access is a full coalesced row;
//  * at 65 536 envs a launch is 1024 waves = ONE wave per SIMD of the chip, so the
//    kernel is a latency chain, not a throughput problem.  It is organised so that a
//    wave makes two memory round trips, not twenty: (1) every unconditional load is
//    issued up front; (2) as soon as the two alive-bitmasks arrive, the live
//    projectile slots are prefetched -- a wave ballot per slot skips slots no lane
//    uses -- and their latency hides under the key / ship / fortress arithmetic;
//  * the per-lane-indexed constants (360-entry cos/sin table, hexagon edges) are
//    staged into LDS once per workgroup; scalar constants are kernel arguments (SGPRs);
//  * the 13 statistics counters are never loaded: events add to them with no-return
//    atomics (executed at the memory side), so a step reads and dirties less;
//  * observations are transposed through LDS and leave as 16-byte coalesced stores;
//  * no dense contraction anywhere on this path: no MFMA.  The bound is HBM traffic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_internal.h"
#include "sf_layout.h"

#define SF_BLOCK 256
#define SF_MAX_MISSILES_D 20.0 /* sf.MAX_MISSILES / sf.MAX_SHELLS as divisors (ENV:124-125) */
#define SF_MPF 8 /* missile slots prefetched into registers; higher slots take the slow loop */
#define SF_SPF 3 /* shell slots prefetched */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// Diagnostic build only (-DSF_STAMPS, tools/stamps.py): shader-clock stamps at phase boundaries,
// with forced waits so each phase owns its memory latency.  The product build has none of it.
#ifdef SF_STAMPS
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

#define SF_PTR(a, name, ctype) \
  (reinterpret_cast<ctype*>((a).state + sfl::offset_per_lane(SF_F_##name) * (a).lanes))

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
};

// what one tick adds to the statistics (SRC/game.hh:29-43); flushed with atomics
// (named scalars, not an array: the compiler merges `if (c) d[5]++; else d[4]++;` into a
// dynamically indexed update, which would push an array into scratch memory)
struct StatDelta {
  int big_hex_deaths = 0, small_hex_deaths = 0, shell_deaths = 0, ship_deaths = 0, resets = 0, destroyed = 0,
      missed = 0, shots = 0, thrusts = 0, lefts = 0, rights = 0, vlner_incs = 0;
  int max_vlner = 0;  // candidate for counter 12 (a running maximum)
};

__device__ __forceinline__ double rad2deg(double a) { return a / M_PI * 180; }  // SRC/vector.cpp:38-40
__device__ __forceinline__ double deg2rad(double a) { return a * M_PI / 180; }  // SRC/vector.cpp:34-36

// Game::reward (SRC/game.cpp:97-102): three float32 adds in this order, points clamped at 0.
__device__ __forceinline__ void score(float amount, float& rew, Lane& L) {
  rew += amount;
  L.raw += amount;
  L.points += amount;
  if (L.points < 0) L.points = 0;
}

// Hexagon::isInside (SRC/hexagon.cpp:36-48); hex -> 6 x (nx, ny, px, py) in LDS.
__device__ __forceinline__ bool hex_inside(const double* hex, double x, double y) {
  bool in = true;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double nx = hex[4 * i + 0], ny = hex[4 * i + 1];
    double dx = x - hex[4 * i + 2], dy = y - hex[4 * i + 3];
    in = in && !(nx * dx + ny * dy < 0);
  }
  return in;
}

// Game::isOutsideGameArea (SRC/game.cpp:129-131)
__device__ __forceinline__ bool outside_area(const SfKernelArgs& a, double x, double y) {
  return x < 0 || x > a.width || y > a.height || y < 0;
}

// Game::resetShip (SRC/game.cpp:133-149).  The accepted (x, y, angle) of the rejection loop over
// libc rand() is a fixed sequence per seed: the host precomputed it (sf_spawn_table) and each
// lane walks it with its own cursor.
__device__ __forceinline__ void spawn_ship(const SfKernelArgs& a, Lane& L) {
  const int16_t* e = a.spawn + 4 * (size_t)(L.cursor & a.spawn_mask);
  L.cursor += 1;
  L.sx = (double)e[0];
  L.sy = (double)e[1];
  L.angle = e[2];
  L.vx = a.start_vx;
  L.vy = a.start_vy;
  L.fl |= SF_FL_SHIP_ALIVE;
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
  L.fort_vuln_t = a.vuln_time;  // :78 adds to a never-initialised member; defined as 0 + 250
  L.mmask = L.smask = 0;
}

__device__ __forceinline__ void kill_ship(Lane& L, StatDelta& S) {  // SRC/game.cpp:274-280
  if (L.fl & SF_FL_SHIP_ALIVE) {
    L.fl &= ~SF_FL_SHIP_ALIVE;
    L.death_t = 0;
    S.ship_deaths += 1;
  }
}

__device__ __forceinline__ void load_lane(const SfKernelArgs& a, long i, Lane& L) {
  L.mmask = SF_PTR(a, missile_mask, uint32_t)[i];  // first: the projectile prefetch waits on these
  L.smask = SF_PTR(a, shell_mask, uint32_t)[i];
  L.sx = SF_PTR(a, ship_x, double)[i];
  L.sy = SF_PTR(a, ship_y, double)[i];
  L.vx = SF_PTR(a, ship_vx, double)[i];
  L.vy = SF_PTR(a, ship_vy, double)[i];
  L.angle = SF_PTR(a, ship_angle, int16_t)[i];
  L.fl = SF_PTR(a, flags, uint8_t)[i];
  L.death_t = SF_PTR(a, ship_death_timer, int32_t)[i];
  L.fire_t = SF_PTR(a, fire_timer, int32_t)[i];
  L.thrust_t = SF_PTR(a, thrust_timer, int32_t)[i];
  L.left_t = SF_PTR(a, left_timer, int32_t)[i];
  L.right_t = SF_PTR(a, right_timer, int32_t)[i];
  L.fort_t = SF_PTR(a, fort_timer, int32_t)[i];
  L.fort_death_t = SF_PTR(a, fort_death_timer, int32_t)[i];
  L.fort_vuln_t = SF_PTR(a, fort_vuln_timer, int32_t)[i];
  L.fort_angle = SF_PTR(a, fort_angle, int16_t)[i];
  L.fort_last = SF_PTR(a, fort_last_angle, int16_t)[i];
  L.points = SF_PTR(a, points, float)[i];
  L.raw = SF_PTR(a, raw_points, float)[i];
  L.vlner = SF_PTR(a, vlner, int32_t)[i];
  L.time = SF_PTR(a, time, int32_t)[i];
  L.prev_vlner = SF_PTR(a, prev_vlner, int32_t)[i];
  L.cursor = SF_PTR(a, spawn_cursor, uint32_t)[i];
}

__device__ __forceinline__ void store_lane(const SfKernelArgs& a, long i, const Lane& L) {
  SF_PTR(a, ship_x, double)[i] = L.sx;
  SF_PTR(a, ship_y, double)[i] = L.sy;
  SF_PTR(a, ship_vx, double)[i] = L.vx;
  SF_PTR(a, ship_vy, double)[i] = L.vy;
  SF_PTR(a, ship_angle, int16_t)[i] = (int16_t)L.angle;
  SF_PTR(a, flags, uint8_t)[i] = (uint8_t)L.fl;
  SF_PTR(a, ship_death_timer, int32_t)[i] = L.death_t;
  SF_PTR(a, fire_timer, int32_t)[i] = L.fire_t;
  SF_PTR(a, thrust_timer, int32_t)[i] = L.thrust_t;
  SF_PTR(a, left_timer, int32_t)[i] = L.left_t;
  SF_PTR(a, right_timer, int32_t)[i] = L.right_t;
  SF_PTR(a, fort_timer, int32_t)[i] = L.fort_t;
  SF_PTR(a, fort_death_timer, int32_t)[i] = L.fort_death_t;
  SF_PTR(a, fort_vuln_timer, int32_t)[i] = L.fort_vuln_t;
  SF_PTR(a, fort_angle, int16_t)[i] = (int16_t)L.fort_angle;
  SF_PTR(a, fort_last_angle, int16_t)[i] = (int16_t)L.fort_last;
  SF_PTR(a, points, float)[i] = L.points;
  SF_PTR(a, raw_points, float)[i] = L.raw;
  SF_PTR(a, vlner, int32_t)[i] = L.vlner;
  SF_PTR(a, time, int32_t)[i] = L.time;
  SF_PTR(a, prev_vlner, int32_t)[i] = L.prev_vlner;
  SF_PTR(a, missile_mask, uint32_t)[i] = L.mmask;
  SF_PTR(a, shell_mask, uint32_t)[i] = L.smask;
}

__device__ __forceinline__ void zero_counters(const SfKernelArgs& a, long i) {
#pragma unroll
  for (int k = 0; k < SF_NSTAT; k++) SF_PTR(a, stats, int32_t)[k * a.lanes + i] = 0;
  SF_PTR(a, ep_return, int32_t)[i] = 0;
  SF_PTR(a, ep_kills, int32_t)[i] = 0;
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
    double dy = L.sy - a.fort_y;
    double ov;
    if (dy == 0)  // on the fortress row the two calls sit on different branch cuts: call it
      ov = atan2(-(a.fort_y - L.sy), a.fort_x - L.sx);
    else
      ov = dy < 0 ? (-M_PI - a_pos) : (M_PI - a_pos);
    double diff = a_vel - ov;
    if (diff > M_PI) diff -= M_PI * 2;
    if (diff < -M_PI) diff += M_PI * 2;
    e.vdir = (L.vx * L.vx + L.vy * L.vy == 0.0) ? 0.0 : rad2deg(diff);
  }
  // fdist, ndist (SRC/game.cpp:310-311): the y term of the reference subtracts the ship from
  // itself, so fdist = sqrt(dx^2 + 0) = |dx|.
  double fdist = fabs(L.sx - a.fort_x);
  e.ndist = -1 + (fdist - a.ndist_a) / a.ndist_b;
  return e;
}

// One observation row (ENV:95-157) written to `o` (an LDS staging row or global memory).
template <typename T>
__device__ __forceinline__ void write_obs(const SfKernelArgs& a, T* o, const Lane& L, const Extras& e) {
  const int n_missiles = __popc(L.mmask);
  const int n_shells = a.real_shell_count ? __popc(L.smask) : n_missiles;  // SRC/pymodule.cpp:131-134
  // ENV:148 reads the vulnerability timer through a getter with undefined behaviour
  // (SRC/pymodule.cpp:44-45); the intended predicate is used.
  const int kill_ready = (L.vlner > 10 && L.fort_vuln_t < a.vuln_time) ? 1 : 0;
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
    f[1] = L.sx / a.pb_width;
    f[2] = L.sy / a.pb_height;
    f[3] = L.vx / 10;
    f[4] = L.vy / 10;
    f[5] = (double)L.angle / 360;
    f[6] = e.aim / 180;
    {
      double m = fmod(e.vdir, 360.0);  // Python float %: result takes the divisor's sign
      if (m != 0) {
        if (m < 0) m += 360.0;
      } else {
        m = 0.0;
      }
      f[7] = m / 360;
    }
    f[8] = e.ndist;
    f[9] = fort_alive ? 1 : 0;
    f[10] = (double)L.fort_angle / 360;
    f[11] = (double)(L.vlner > 10 ? L.vlner : 10) / 10;  // ENV:122 max(), as written
    f[12] = kill_ready;
    f[13] = (double)n_missiles / SF_MAX_MISSILES_D;
    f[14] = (double)n_shells / SF_MAX_MISSILES_D;
#pragma unroll
    for (int k = 0; k < 4; k++) f[15 + k] = (double)timers[k] / a.max_ticks;
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

// The block's observation rows sit in LDS as [lane][obs_dim]; global memory wants exactly the
// same order ([N, obs_dim] row-major), so the block's rows form one contiguous span: copy it
// with 16-byte lanes-consecutive stores instead of obs_dim strided 4-byte stores per lane.
template <typename T>
__device__ __forceinline__ void flush_obs_block(const SfKernelArgs& a, const T* stage, T* obs, bool vec_ok) {
  const long base = (long)blockIdx.x * SF_BLOCK;
  long rows = (long)a.n_envs - base;
  if (rows > SF_BLOCK) rows = SF_BLOCK;
  if (rows <= 0) return;
  const int total = (int)rows * a.obs_dim;
  T* dst = obs + base * a.obs_dim;
  constexpr int V = 16 / (int)sizeof(T);
  int done_elems = 0;
  if (vec_ok) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int nvec = total / V;
    for (int v = threadIdx.x; v < nvec; v += SF_BLOCK)
      reinterpret_cast<vec_t*>(dst)[v] = reinterpret_cast<const vec_t*>(stage)[v];
    done_elems = nvec * V;
  }
  for (int t = done_elems + threadIdx.x; t < total; t += SF_BLOCK) dst[t] = stage[t];
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// VecEnv.reset(): a brand-new Game in every lane (ENV:163-178).  first != 0 additionally
// initialises what SSF_Env.__init__ sets once (prev_vlner, ENV:92) and the spawn cursors.
__global__ __launch_bounds__(SF_BLOCK) void sf_reset_kernel(SfKernelArgs a, int first, unsigned cursor0,
                                                           unsigned cursor_stride, void* obs) {
  const long i = (long)blockIdx.x * SF_BLOCK + threadIdx.x;
  Lane L;
  if (first) {
    L.prev_vlner = 0;
    L.cursor = cursor0 + cursor_stride * (unsigned)i;
  } else {
    L.prev_vlner = SF_PTR(a, prev_vlner, int32_t)[i];
    L.cursor = SF_PTR(a, spawn_cursor, uint32_t)[i];
  }
  new_game(a, L);
  store_lane(a, i, L);
  SF_PTR(a, spawn_cursor, uint32_t)[i] = L.cursor;
  zero_counters(a, i);
  if (obs != nullptr && i < a.n_envs && a.obs_type != 3) {
    // the reference's extras are stale heap until the first tick; defined as computeExtra(spawn)
    Extras e = compute_extras(a, L, atan2(L.sy - a.fort_y, L.sx - a.fort_x), atan2(L.vy, L.vx));
    if (a.obs_f64)
      write_obs<double>(a, (double*)obs + (size_t)i * a.obs_dim, L, e);
    else
      write_obs<float>(a, (float*)obs + (size_t)i * a.obs_dim, L, e);
  }
}

// ---------------------------------------------------------------------------------------------
template <bool AUTOTURN>
__global__ __launch_bounds__(SF_BLOCK) void sf_step_kernel(SfKernelArgs a, const void* actions, int act_type,
                                                          void* obs, int obs_vec_ok, int32_t* reward_out,
                                                          uint8_t* done_out, uint8_t* info_out) {
  extern __shared__ double lds[];  // [SF_LDS_DOUBLES] constants, then the obs staging rows
  const int tid = threadIdx.x;
  const long i = (long)blockIdx.x * SF_BLOCK + tid;
  const bool real = i < a.n_envs;  // lanes in [n_envs, lanes) are padding: they run NOOPs
#ifdef SF_STAMPS
  unsigned long long stamp_[16];
  stamp_[12] = __builtin_amdgcn_s_memrealtime();
#endif
  SF_STAMP(0, false);

  // ================= round trip 1: every unconditional load =================
  int act = 0;
  if (real) {  // ENV:211-212
    if (act_type == 8)
      act = (int)((const long long*)actions)[i];
    else if (act_type == 4)
      act = ((const int*)actions)[i];
    else
      act = ((const unsigned char*)actions)[i];
  }
  Lane L;
  load_lane(a, i, L);
  // the constant block: SF_LDS_DOUBLES == 3 * SF_BLOCK
  const double c0 = a.consts[tid], c1 = a.consts[tid + SF_BLOCK], c2 = a.consts[tid + 2 * SF_BLOCK];
  SF_STAMP(1, false);
  SF_STAMP(2, true);

  // ================= round trip 2: live projectile slots, predicated by the alive masks ======
  double mx[SF_MPF], my[SF_MPF];
  int mang[SF_MPF];
  double shx[SF_SPF], shy[SF_SPF], shvx[SF_SPF], shvy[SF_SPF];
  {
    const double* gx = SF_PTR(a, missile_x, double);
    const double* gy = SF_PTR(a, missile_y, double);
    const int16_t* ga = SF_PTR(a, missile_angle, int16_t);
#pragma unroll
    for (int s = 0; s < SF_MPF; s++) {
      const bool live = (L.mmask >> s) & 1u;
      mx[s] = 0;
      my[s] = 0;
      mang[s] = 0;
      if (__ballot(live) != 0ull) {  // some lane of this wave uses slot s
        if (live) {
          const long idx = (long)s * a.lanes + i;
          mx[s] = gx[idx];
          my[s] = gy[idx];
          mang[s] = ga[idx];
        }
      }
    }
    const double* hx = SF_PTR(a, shell_x, double);
    const double* hy = SF_PTR(a, shell_y, double);
    const double* hvx = SF_PTR(a, shell_vx, double);
    const double* hvy = SF_PTR(a, shell_vy, double);
#pragma unroll
    for (int s = 0; s < SF_SPF; s++) {
      const bool live = (L.smask >> s) & 1u;
      shx[s] = shy[s] = shvx[s] = shvy[s] = 0;
      if (__ballot(live) != 0ull) {
        if (live) {
          const long idx = (long)s * a.lanes + i;
          shx[s] = hx[idx];
          shy[s] = hy[idx];
          shvx[s] = hvx[idx];
          shvy[s] = hvy[idx];
        }
      }
    }
  }

  // constants -> LDS (the loads were issued in round trip 1)
  lds[tid] = c0;
  lds[tid + SF_BLOCK] = c1;
  lds[tid + 2 * SF_BLOCK] = c2;
  __syncthreads();
  const double* trig = lds + SF_LDS_TRIG;
  SF_STAMP(3, false);

  if (act < 0 || act >= a.n_actions) {
    atomicAdd(&a.acc[8], 1ull);  // reference: IndexError; here NOOP + counted (sf_check_actions)
    act = 0;
  }
  const unsigned keys = (unsigned)(a.action_keys >> (4 * act)) & 0xFu;

  StatDelta S;

  // ================= Game::stepOneTick (SRC/game.cpp:473-485) =================
  float rew = 0;        // mReward = 0
  L.time += a.tick_ms;  // updateTime
  const unsigned cursor0 = L.cursor;

  // ---- processKeyState (SRC/game.cpp:218-272); the wrapper sends FIRE, THRUST, (LEFT, RIGHT)
  //      press-or-release every step (ENV:213-229), so each key is one edge test.
  int new_m_slot = -1;
  {
    const bool k_fire = keys & 1u, k_thrust = keys & 2u;
    if (k_fire && !(L.fl & SF_FL_FIRE)) {
      // fireMissile (SRC/game.cpp:175-192): before this tick's turn and move
      if (L.fl & SF_FL_SHIP_ALIVE) {
        int slot = __ffs(~L.mmask) - 1;
        if (slot < SF_NSLOT) {
          new_m_slot = slot;
          L.mmask |= 1u << slot;
          score(-a.missile_penalty, rew, L);
        }
      }
      L.fl |= SF_FL_FIRE;
      L.fire_t = 0;
      S.shots += 1;
    } else if (!k_fire && (L.fl & SF_FL_FIRE)) {
      L.fl &= ~SF_FL_FIRE;
      L.fire_t = 0;
    }
    if (k_thrust && !(L.fl & SF_FL_THRUST)) {
      L.fl |= SF_FL_THRUST;
      L.thrust_t = 0;
      S.thrusts += 1;
    } else if (!k_thrust && (L.fl & SF_FL_THRUST)) {
      L.fl &= ~SF_FL_THRUST;
      L.thrust_t = 0;
    }
    if (!AUTOTURN) {
      const bool k_left = keys & 4u, k_right = keys & 8u;
      if (k_left && !(L.fl & SF_FL_LEFT)) {
        L.fl |= SF_FL_LEFT;
        L.left_t = 0;
        S.lefts += 1;
      } else if (!k_left && (L.fl & SF_FL_LEFT)) {
        L.fl &= ~SF_FL_LEFT;
        L.left_t = 0;
      }
      if (k_right && !(L.fl & SF_FL_RIGHT)) {
        L.fl |= SF_FL_RIGHT;
        L.right_t = 0;
        S.rights += 1;
      } else if (!k_right && (L.fl & SF_FL_RIGHT)) {
        L.fl &= ~SF_FL_RIGHT;
        L.right_t = 0;
      }
    }
  }
  // the missile created above starts at the ship's pre-move position and heading
  const double new_m_x = L.sx, new_m_y = L.sy;
  const int new_m_angle = L.angle;

  // ---- monitorShipRespawn (SRC/game.cpp:151-157)
  if (!(L.fl & SF_FL_SHIP_ALIVE) && L.death_t >= a.explode_duration) {
    spawn_ship(a, L);
    L.fort_t = 0;
  }

  // ---- updateShip (SRC/game.cpp:314-351)
  if (L.fl & SF_FL_SHIP_ALIVE) {
    if (AUTOTURN) {
      // stdAngle(ceil(angleTo(ship, fortress)))  (SRC/vector.cpp:42-52)
      double t = atan2(a.fort_y - L.sy, a.fort_x - L.sx);
      if (t < 0) t += M_PI * 2;
      double c = ceil(rad2deg(t));  // in [0, 360]
      int ia = (int)c;
      if (ia >= 360) ia -= 360;  // fmod(360, 360)
      L.angle = ia;
    } else {
      const bool left = L.fl & SF_FL_LEFT, right = L.fl & SF_FL_RIGHT;
      if (left && !right) {  // TURN_LEFT: stdAngle(angle - turnSpeed)
        L.angle -= a.turn_speed;
        if (L.angle < 0) L.angle += 360;
      } else if (right && !left) {  // TURN_RIGHT
        L.angle += a.turn_speed;
        if (L.angle >= 360) L.angle -= 360;
      }
    }
    if (L.fl & SF_FL_THRUST) {
      L.vx += a.ship_accel * trig[2 * L.angle];
      L.vy += a.ship_accel * trig[2 * L.angle + 1];
    }
    L.sx += L.vx;
    L.sy += L.vy;
    // `if (!big.isInside) ... else if (small.isInside) ...` (:337-349), counters branch-free
    const int out_big = !hex_inside(lds + SF_LDS_BIGHEX, L.sx, L.sy);
    const int in_small = !out_big && hex_inside(lds + SF_LDS_SMALLHEX, L.sx, L.sy);
    if (out_big | in_small) {
      kill_ship(L, S);
      score(-a.death_penalty, rew, L);
    }
    S.big_hex_deaths += out_big;
    S.small_hex_deaths += in_small;
  }

  // the two bearings the rest of the tick and the observation need, side by side (ILP)
  double a_pos = atan2(L.sy - a.fort_y, L.sx - a.fort_x);
  double a_vel = atan2(L.vy, L.vx);

  // ---- updateFortress (SRC/game.cpp:194-216)
  int new_s_slot = -1;
  double new_s_vx = 0, new_s_vy = 0;
  {
    double ats = rad2deg(a_pos);  // stdAngle: in [-180, 180], only the sign fix applies
    if (ats < 0) ats += 360;
    if (!(L.fl & SF_FL_FORT_ALIVE) && L.fort_death_t > 1000) {
      L.fort_t = 0;
      L.fl |= SF_FL_FORT_ALIVE;
    }
    if (L.fl & SF_FL_SHIP_ALIVE) {
      double q = ceil(ats / a.sector_size) * a.sector_size;  // in [0, 360]
      int fa = (int)q;
      if (fa >= 360) fa -= 360;
      L.fort_angle = fa;
      if (fa != L.fort_last) {
        L.fort_last = fa;
        L.fort_t = 0;
      }
      if (L.fort_t >= a.lock_time && (L.fl & SF_FL_FORT_ALIVE)) {
        // fireShell (SRC/game.cpp:159-173): a non-integer heading, so real sin/cos
        int slot = __ffs(~L.smask) - 1;
        if (slot < SF_NSLOT) {
          new_s_slot = slot;
          L.smask |= 1u << slot;
          double r = deg2rad(ats);
          new_s_vx = a.shell_speed * cos(r);
          new_s_vy = a.shell_speed * sin(r);
        }
        L.fort_t = 0;
      }
    }
  }

  SF_STAMP(4, false);
  SF_STAMP(5, true);
  // ---- updateShells (SRC/game.cpp:404-423), slot order
  {
    double* px = SF_PTR(a, shell_x, double);
    double* py = SF_PTR(a, shell_y, double);
    double* pvx = SF_PTR(a, shell_vx, double);
    double* pvy = SF_PTR(a, shell_vy, double);
    auto shell_step = [&](int s, double x, double y, double vx, double vy, bool isnew) __attribute__((always_inline)) {
      const long idx = (long)s * a.lanes + i;
      if (isnew) {
        pvx[idx] = vx;
        pvy[idx] = vy;
      }
      x += vx;
      y += vy;
      bool dead = false;
      if (L.fl & SF_FL_SHIP_ALIVE) {
        // Object::collided (SRC/object.cpp:12-15): sqrt(dx^2+dy^2) <= r.  With a correctly
        // rounded sqrt and r an integer, RN(sqrt(s)) <= r  <=>  s <= r^2 (r^2 is exactly
        // representable and the next double above r^2 has a root that rounds above r).
        double dx = x - L.sx, dy = y - L.sy;
        if (dx * dx + dy * dy <= a.shell_hit_r2) {
          dead = true;
          kill_ship(L, S);
          score(-a.death_penalty, rew, L);
          S.shell_deaths += 1;
        }
      }
      if (!dead && outside_area(a, x, y)) dead = true;
      if (dead) {
        L.smask &= ~(1u << s);
      } else {
        px[idx] = x;
        py[idx] = y;
      }
    };
#pragma unroll
    for (int s = 0; s < SF_SPF; s++) {
      const bool live = (L.smask >> s) & 1u;
      if (__ballot(live) == 0ull) continue;
      if (live) {
        const bool isnew = (s == new_s_slot);
        shell_step(s, isnew ? a.fort_x : shx[s], isnew ? a.fort_y : shy[s], isnew ? new_s_vx : shvx[s],
                   isnew ? new_s_vy : shvy[s], isnew);
      }
    }
    if (__ballot((L.smask >> SF_SPF) != 0u) != 0ull) {  // rare: more than SF_SPF shells in some lane
      for (int s = SF_SPF; s < SF_NSLOT; s++) {
        const bool live = (L.smask >> s) & 1u;
        if (__ballot(live) == 0ull) continue;
        if (live) {
          const long idx = (long)s * a.lanes + i;
          if (s == new_s_slot)
            shell_step(s, a.fort_x, a.fort_y, new_s_vx, new_s_vy, true);
          else
            shell_step(s, px[idx], py[idx], pvx[idx], pvy[idx], false);
        }
      }
    }
  }

  // ---- updateMissiles (SRC/game.cpp:353-402), slot order
  {
    double* px = SF_PTR(a, missile_x, double);
    double* py = SF_PTR(a, missile_y, double);
    int16_t* pa = SF_PTR(a, missile_angle, int16_t);
    auto missile_step = [&](int s, double x, double y, int ang, bool isnew) __attribute__((always_inline)) {
      const long idx = (long)s * a.lanes + i;
      if (isnew) pa[idx] = (int16_t)ang;
      // velocity = missileSpeed * (cos, sin)(deg2rad(angle)) with an integer angle: table
      x += a.missile_speed * trig[2 * ang];
      y += a.missile_speed * trig[2 * ang + 1];
      double dx = x - a.fort_x, dy = y - a.fort_y;
      bool dead = false;
      if (dx * dx + dy * dy <= a.missile_hit_r2) {  // collided(mFortress), see the shell note
        dead = true;
        if (L.fl & SF_FL_FORT_ALIVE) {
          if (L.fort_vuln_t >= a.vuln_time) {
            L.vlner += 1;
            S.vlner_incs += 1;
            if (L.vlner > S.max_vlner) S.max_vlner = L.vlner;
          } else {
            const int destroy = L.vlner >= a.vuln_threshold + 1;
            if (destroy) {
              L.fl &= ~SF_FL_FORT_ALIVE;
              L.fort_death_t = 0;
              score(a.destroy_reward, rew, L);
            }
            // branch-free on purpose: `if (c) destroyed++; else resets++;` gets merged into a
            // store through a selected address, which forces the counters into scratch memory
            S.destroyed += destroy;
            S.resets += 1 - destroy;
            L.vlner = 0;
          }
          L.fort_vuln_t = 0;
        }
      } else if (outside_area(a, x, y)) {
        dead = true;
        score(-a.miss_penalty, rew, L);
        S.missed += 1;
      }
      if (dead) {
        L.mmask &= ~(1u << s);
      } else {
        px[idx] = x;
        py[idx] = y;
      }
    };
#pragma unroll
    for (int s = 0; s < SF_MPF; s++) {
      const bool live = (L.mmask >> s) & 1u;
      if (__ballot(live) == 0ull) continue;
      if (live) {
        const bool isnew = (s == new_m_slot);
        missile_step(s, isnew ? new_m_x : mx[s], isnew ? new_m_y : my[s], isnew ? new_m_angle : mang[s], isnew);
      }
    }
    if (__ballot((L.mmask >> SF_MPF) != 0u) != 0ull) {  // rare: a lane with more than SF_MPF missiles
      for (int s = SF_MPF; s < SF_NSLOT; s++) {
        const bool live = (L.mmask >> s) & 1u;
        if (__ballot(live) == 0ull) continue;
        if (live) {
          const long idx = (long)s * a.lanes + i;
          if (s == new_m_slot)
            missile_step(s, new_m_x, new_m_y, new_m_angle, true);
          else
            missile_step(s, px[idx], py[idx], pa[idx], false);
        }
      }
    }
  }

  SF_STAMP(6, false);
  // ---- stepTimers (SRC/game.cpp:425-451)
  L.fort_t += a.tick_ms;
  L.fort_death_t += a.tick_ms;
  L.fort_vuln_t += a.tick_ms;
  L.death_t += a.tick_ms;
  L.fire_t += (L.fl & SF_FL_FIRE) ? 1 : -1;
  L.thrust_t += (L.fl & SF_FL_THRUST) ? 1 : -1;
  L.left_t += (L.fl & SF_FL_LEFT) ? 1 : -1;
  L.right_t += (L.fl & SF_FL_RIGHT) ? 1 : -1;

  int r = (int)rew;  // `return mReward` through `int stepOneTick` (SRC/game.hh:138): truncation

  // ================= SSF_Env.step epilogue (ENV:233-253) =================
  const int fort_kill = r > 0;
  if (a.shaped) {
    const int vlner_change = L.vlner - L.prev_vlner;
    if (L.vlner <= 10 && !fort_kill) r += vlner_change;
    r = r > 1 ? 1 : (r < -1 ? -1 : r);
    r = r + 2 * fort_kill;
    L.prev_vlner = L.vlner;
  }
  const int done = L.time >= a.game_time;  // Game::isGameOver (SRC/game.cpp:487-489)

  // ================= statistics and the vec-env worker's auto-reset (rl/train.py:80-88) ======
  int32_t* gstats = SF_PTR(a, stats, int32_t);
  if (done && a.auto_reset) {
    // episode totals = what previous launches accumulated + this tick's share
    const int ep_ret = SF_PTR(a, ep_return, int32_t)[i] + r;
    const int ep_kil = SF_PTR(a, ep_kills, int32_t)[i] + fort_kill;
    const int deaths = gstats[SF_ST_SHIP_DEATHS * a.lanes + i] + S.ship_deaths;
    const int shots = gstats[SF_ST_SHOTS * a.lanes + i] + S.shots;
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
    new_game(a, L);
    zero_counters(a, i);
    a_pos = atan2(L.sy - a.fort_y, L.sx - a.fort_x);
    a_vel = atan2(L.vy, L.vx);
  } else {
    // no-return atomics, executed at the memory side: the counters are never loaded
#define SF_FLUSH(idx, v)                                        \
  if (__ballot((v) != 0) != 0ull) {                             \
    if ((v) != 0) atomicAdd(&gstats[(idx)*a.lanes + i], (v));   \
  }
    SF_FLUSH(SF_ST_BIG_HEX_DEATHS, S.big_hex_deaths)
    SF_FLUSH(SF_ST_SMALL_HEX_DEATHS, S.small_hex_deaths)
    SF_FLUSH(SF_ST_SHELL_DEATHS, S.shell_deaths)
    SF_FLUSH(SF_ST_SHIP_DEATHS, S.ship_deaths)
    SF_FLUSH(SF_ST_RESETS, S.resets)
    SF_FLUSH(SF_ST_DESTROYED, S.destroyed)
    SF_FLUSH(SF_ST_MISSED, S.missed)
    SF_FLUSH(SF_ST_SHOTS, S.shots)
    SF_FLUSH(SF_ST_THRUSTS, S.thrusts)
    SF_FLUSH(SF_ST_LEFTS, S.lefts)
    SF_FLUSH(SF_ST_RIGHTS, S.rights)
    SF_FLUSH(SF_ST_VLNER_INCS, S.vlner_incs)
#undef SF_FLUSH
    if (__ballot(S.max_vlner != 0) != 0ull) {
      if (S.max_vlner != 0) atomicMax(&gstats[SF_ST_MAX_VLNER * a.lanes + i], S.max_vlner);
    }
    if (__ballot(r != 0) != 0ull) {
      if (r != 0) atomicAdd(&SF_PTR(a, ep_return, int32_t)[i], r);
    }
    if (__ballot(fort_kill != 0) != 0ull) {
      if (fort_kill) atomicAdd(&SF_PTR(a, ep_kills, int32_t)[i], 1);
    }
  }

  store_lane(a, i, L);
  if (__ballot(L.cursor != cursor0) != 0ull) SF_PTR(a, spawn_cursor, uint32_t)[i] = L.cursor;

  if (real) {
    if (reward_out) reward_out[i] = r;
    if (done_out) done_out[i] = (uint8_t)done;
    if (info_out) info_out[i] = (uint8_t)fort_kill;
  }
  SF_STAMP(7, false);
  if (obs != nullptr && a.obs_type != 3) {  // uniform across the grid
    const Extras e = compute_extras(a, L, a_pos, a_vel);
    if (a.obs_f64) {
      double* stage = lds + SF_LDS_DOUBLES;
      write_obs<double>(a, stage + tid * a.obs_dim, L, e);
      __syncthreads();
      flush_obs_block<double>(a, stage, (double*)obs, obs_vec_ok);
    } else {
      float* stage = reinterpret_cast<float*>(lds + SF_LDS_DOUBLES);
      write_obs<float>(a, stage + tid * a.obs_dim, L, e);
      __syncthreads();
      flush_obs_block<float>(a, stage, (float*)obs, obs_vec_ok);
    }
  }
  SF_STAMP(8, false);
  SF_STAMP(9, true);
#ifdef SF_STAMPS
  stamp_[13] = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (a.dbg != nullptr && (tid & 63) == 0) {
    unsigned long long* d = a.dbg + ((size_t)blockIdx.x * (SF_BLOCK / 64) + (tid >> 6)) * 16;
#pragma unroll
    for (int k = 0; k < 14; k++) d[k] = (k < 10 || k >= 12) ? stamp_[k] : 0ull;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// launchers (called by sf_capi.cpp)

hipError_t sf_launch_reset(const SfKernelArgs& a, int first, unsigned cursor0, unsigned stride, void* obs,
                           hipStream_t stream) {
  const unsigned grid = (unsigned)(a.lanes / SF_BLOCK);
  hipLaunchKernelGGL(sf_reset_kernel, dim3(grid), dim3(SF_BLOCK), 0, stream, a, first, cursor0, stride, obs);
  return hipGetLastError();
}

hipError_t sf_launch_step(const SfKernelArgs& a, bool autoturn, const void* actions, int act_type, void* obs,
                          int32_t* reward, uint8_t* done, uint8_t* info, hipStream_t stream) {
  const unsigned grid = (unsigned)(a.lanes / SF_BLOCK);
  const size_t elem = a.obs_f64 ? sizeof(double) : sizeof(float);
  const size_t lds_bytes = SF_LDS_DOUBLES * sizeof(double) + (size_t)SF_BLOCK * a.obs_dim * elem;
  const int vec_ok = ((uintptr_t)obs & 15u) == 0;
  if (autoturn)
    hipLaunchKernelGGL(sf_step_kernel<true>, dim3(grid), dim3(SF_BLOCK), lds_bytes, stream, a, actions, act_type,
                       obs, vec_ok, reward, done, info);
  else
    hipLaunchKernelGGL(sf_step_kernel<false>, dim3(grid), dim3(SF_BLOCK), lds_bytes, stream, a, actions, act_type,
                       obs, vec_ok, reward, done, info);
  return hipGetLastError();
}
