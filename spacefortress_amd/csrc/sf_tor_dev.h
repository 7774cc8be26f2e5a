// sf_tor_dev.h -- cairo's coverage of the frame's small objects, for one WAVE: the lane arrangement of sf_tor.h's arithmetic.
// Device code only; used by sf_render.hip (the default geometry's frame kernel) and sf_render_generic.hip (any geometry).
//
// One call rasterises up to kMaxQuads convex quads belonging to up to kMaxObjs objects (an object = one cairo_stroke of the
// reference: the ship's three lines, a missile's three, a shell's four, one arc of an explosion, its circle's sixteen
// pieces) and composites each object, in order, onto the 8-bit surface in LDS:
//
//   records   the owner lane of a quad prepares it (quad_scan: the edges' A + B s, left / right, sub-row range) and leaves a
//             record in LDS; the object's first lane its box, sub-row range and accumulator
//   rows      a lane per (object, pixel row): can_do_full_row -- no vertex strictly inside the row, the edges keep their order --
//             and, for a row taken whole, cairo's trapezoid areas of every source's outer edges (row_edge), added to the
//             object's accumulator
//   sub-rows  a lane per (object, sub-row) of the other rows: every quad's span [L, R), then every SOURCE of the object -- a
//             quad alone (+), two that can overlap (-), three (+): the union of the object's quads, which is what one
//             cairo_stroke with non-zero winding covers -- adds 2 * (length inside the pixel) to the pixels it touches
//   pixels    a lane per pixel of the object's box: alpha = (17 c + 256) >> 9, lerp into the surface; the accumulator is
//             cleared for the next call
//
// Accumulators are 16 bits per pixel, two to an LDS word, changed by ds_add_u32: a pixel's sum never exceeds 4 * 7680 and
// the negative terms of a sub-row come after its positive ones (same wave, LDS in order), so no half ever borrows.
#pragma once
#include <hip/hip_runtime.h>

#include "sf_tor.h"

namespace sftd {

constexpr int kMaxQuads = 16, kMaxObjs = 16;  // (the default geometry's kernel; the general one takes kMaxQuadsBig)
constexpr int kMaxQuadsBig = 32;
constexpr int kRecWords = 28;  // A, B of four edges (16), s0, s1, masks, spare (4), vertices x[4], y[4] (8)
constexpr int kObjWords = 8;   // box x0 | y0 << 8 | w << 16 | h << 24; S0; nsub; acc base (pixels); first quad | nq << 8 | kind << 16 | grey << 24; row0 | nrows << 8; modes; -
constexpr int kAccPixels = 640;
constexpr int kLdsWords = kMaxQuads * kRecWords + kMaxObjs * kObjWords + kAccPixels / 2;
constexpr int kLdsWordsBig = kMaxQuadsBig * kRecWords + kMaxObjs * kObjWords + kAccPixels / 2;

// what may overlap inside an object, by kind: up to four sets of quad SLOTS (bit masks over the object's first four slots;
// the circle's seams use slots 0 = piece 0, 1 = piece 7, 2 = piece 8, 3 = piece 15) with the sign of inclusion-exclusion
// (a circle's kind carries the number of pieces of its first half in bits 8..15: kKindRing | m0 << 8 -- eight in the default geometry)
enum { kKindLines3 = 0, kKindShell = 1, kKindSingle = 2, kKindRing = 3, kKindFort = 4 };
__device__ __forceinline__ unsigned multi_sources(int kind) {  // four nibbles: member masks; signs: 2 members -, 3 members +
  kind &= 255;
  return kind == kKindLines3 ? 0x7653u      // {0,1} {0,2} {1,2} {0,1,2}
         : kind == kKindShell ? 0x9C63u     // {0,1} {1,2} {2,3} {3,0}
         : kind == kKindRing ? 0x0096u      // {1,2} = pieces 7 | 8, {0,3} = pieces 0 | 15
         : kind == kKindFort ? 0x05C6u      // {1,2} {2,3} {0,2}: the two corners and the bar through the upright
                             : 0u;
}

struct Ctx {
  uint32_t* lds;   // kLdsWords (maxq = kMaxQuads) or kLdsWordsBig (maxq = kMaxQuadsBig)
  uint8_t* fb;     // the surface
  int W, H, lane;
  int maxq = kMaxQuads;
  int acc_at = -1;  // where the accumulators start (words); -1: right behind the objects.  A kernel that uses BOTH arrangements
                    // gives them one accumulator (each call leaves it zero; what lies below it is scratch of either)
  __device__ __forceinline__ uint32_t* rec(int q) const { return lds + q * kRecWords; }
  __device__ __forceinline__ uint32_t* obj(int o) const { return lds + maxq * kRecWords + o * kObjWords; }
  __device__ __forceinline__ uint32_t* acc() const { return lds + (acc_at >= 0 ? acc_at : maxq * kRecWords + kMaxObjs * kObjWords); }
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};

struct RecView {  // a quad's record, read back from LDS
  const uint32_t* r;
  __device__ __forceinline__ sft::EdgeAB edge(int e) const {
    const double2 v = *reinterpret_cast<const double2*>(r + 4 * e);
    return sft::EdgeAB{v.x, v.y};
  }
  __device__ __forceinline__ int s0() const { return (int)r[16]; }
  __device__ __forceinline__ int s1() const { return (int)r[17]; }
  __device__ __forceinline__ unsigned left() const { return r[18] & 15u; }
  __device__ __forceinline__ unsigned horiz() const { return (r[18] >> 4) & 15u; }
  __device__ __forceinline__ bool has_out() const { return (r[18] >> 8) & 1u; }
  __device__ __forceinline__ int vx(int k) const { return (int)r[20 + k]; }
  __device__ __forceinline__ int vy(int k) const { return (int)r[24 + k]; }
};

__device__ __forceinline__ void acc_add(uint32_t* acc, int pixel, int v) {
  __hip_atomic_fetch_add(acc + (pixel >> 1), (uint32_t)v << (16 * (pixel & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the span of quad `rv` in sub-row s (the caller has checked s0 <= s < s1)
__device__ __forceinline__ void rec_interval(const RecView& rv, int s, int xmax, int* L, int* R) {
  int l = sft::kCellMin, r = sft::kCellMax;
  const unsigned left = rv.left(), horiz = rv.horiz();
  int c[4];
#pragma unroll
  for (int e = 0; e < 4; e++) c[e] = sft::edge_cell(rv.edge(e), s);
  if (rv.has_out()) {  // (a stroke across the surface's left or right border: rare, and then the slow way)
    sft::Quad q;
#pragma unroll
    for (int k = 0; k < 4; k++) { q.x[k] = rv.vx(k); q.y[k] = rv.vy(k); }
    const sft::QuadScan qs = sft::quad_scan(q, xmax);
#pragma unroll
    for (int e = 0; e < 4; e++)
      if (s >= qs.out_s0[e] && s < qs.out_s1[e]) c[e] = qs.out_x[e];
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    if ((horiz >> e) & 1u) continue;
    if ((left >> e) & 1u) l = c[e] > l ? c[e] : l;
    else r = c[e] < r ? c[e] : r;
  }
  *L = l; *R = r;
}

// add `sign` * 2 * (length of [L, R) inside each pixel) to row `arow` (an offset into the object's accumulator) of a box
// that starts at pixel column bx0 and is bw wide
__device__ __forceinline__ void add_span(uint32_t* acc, int arow, int bx0, int bw, int L, int R, int sign) {
  if (R <= L) return;
  int px = L >> 8;
  px = px < bx0 ? bx0 : px;
  const int last = min((R - 1) >> 8, bx0 + bw - 1);
  for (; px <= last; px++) {
    const int lo = max(L, px << 8), hi = min(R, (px + 1) << 8);
    if (hi > lo) acc_add(acc, arow + px - bx0, sign * 2 * (hi - lo));
  }
}

// ---- the call -------------------------------------------------------------------------------------------------------------------
// Lane l holds quad l of the call (`valid`), lanes of one object are consecutive and `obj0` is the object's first lane; `kind`
// and `grey` are the object's (the same in all its lanes).  At most kMaxQuads valid lanes and kMaxObjs objects per call, and
// the objects' boxes must fit kAccPixels: the caller chunks (wireframes: up to five objects; an explosion's ring: twelve arcs).
// the slot (0..3) a quad has in its object's overlap sets, or -1: the quad's index for lines; a circle's seam pieces
__device__ __forceinline__ int slot_of(int okind, int k, int nq, int m0) {
  return okind == kKindRing ? (k == 0 ? 0 : (k == m0 - 1 ? 1 : (k == m0 ? 2 : (k == nq - 1 ? 3 : -1)))) : (k < 4 ? k : -1);
}

// MAXACT: how many quads of one object a pixel row taken whole may hold (4: lines; 8: a circle at a large scale -- more and the
// row is sampled in sub-rows instead, which cairo does not do: the general kernel's circles stay below)
#ifndef SFTD_STOP
#define SFTD_STOP 9 /* diagnostic builds: leave raster() behind phase N (0: at once, 1 records + boxes, 2 rows, 3 sub-rows) */
#endif
template <int MAXACT = 4>
__device__ __forceinline__ void raster(const Ctx& C, const sft::Quad& mine, bool valid, int obj0, int kind, int grey) {
  if (SFTD_STOP == 0) return;
  const int lane = C.lane, XM = C.W * 256;
  uint32_t* const acc = C.acc();
  // ---- records
  const unsigned long long vmask = __ballot(valid);
  if (!vmask) return;
  const int qi = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vmask, 0u));  // my quad's index
  const bool leader = valid && lane == obj0;
  const unsigned long long lmask = __ballot(leader);
  const int oi = (int)__popcll(lmask & ((2ull << obj0) - 1ull)) - 1;  // my object's index (obj0 <= lane: counts leaders up to obj0)
  const int nobj = (int)__popcll(lmask);
  sft::QuadScan qs;
  int minx = 1 << 30, maxx = -(1 << 30);
  if (valid) {
    qs = sft::quad_scan(mine, XM);
    uint32_t* r = C.rec(qi);
    bool any_out = false;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      *reinterpret_cast<double2*>(r + 4 * e) = double2{qs.e[e].A, qs.e[e].B};
      any_out |= qs.out_s1[e] > qs.out_s0[e];
      r[20 + e] = (uint32_t)mine.x[e];
      r[24 + e] = (uint32_t)mine.y[e];
      minx = min(minx, mine.x[e]);
      maxx = max(maxx, mine.x[e]);
    }
    r[16] = (uint32_t)qs.s0; r[17] = (uint32_t)qs.s1;
    r[18] = qs.left | (qs.horiz << 4) | (any_out ? 256u : 0u);
    r[19] = 0u;
  }
  if (leader) {
    uint32_t* o = C.obj(oi);
    o[0] = 0x7fffffffu; o[1] = 0x7fffffffu; o[2] = 0x80000000u; o[3] = 0x80000000u;  // min x, min s, max x, max s (as ints)
    o[4] = (uint32_t)qi | ((uint32_t)(kind & 255) << 16) | ((uint32_t)grey << 24);
    o[5] = 0u; o[6] = 0u; o[7] = (uint32_t)(kind >> 8);  // (a circle's first half: kept until the sub-rows' start replaces o[7])
  }
  C.sync();
  if (valid) {
    int* o = reinterpret_cast<int*>(C.obj(oi));
    __hip_atomic_fetch_min(o + 0, minx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(o + 1, qs.s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(o + 2, maxx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(o + 3, qs.s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(C.obj(oi) + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // the object's quads
  }
  C.sync();
  // ---- the objects' boxes, in one lane each (lane o < nobj).  The accumulators hold kAccPixels pixels and a row's mode is a bit
  // of one word: objects that are larger (another geometry's close-up: a ship of forty rows, an arc sixty pixels out) are taken in
  // WINDOWS of pixel rows, each through all the passes below -- a row's mode, its whole-row areas and its sub-rows depend on nothing
  // outside the row, so the windows add up to the same picture.
  int fx0 = 0, fw = 0, fs0 = 0, fs1 = 0, m0_mine = 0;
  if (lane < nobj) {
    uint32_t* o = C.obj(lane);
    const int* oi_ = reinterpret_cast<const int*>(o);
    const int x0 = max(oi_[0] >> 8, 0), x1 = min((oi_[2] + 255) >> 8, C.W);
    const int s0 = max(oi_[1], 0), s1 = min(oi_[3], C.H * sft::kGridY);
    if (x1 > x0 && s1 > s0) { fx0 = x0; fw = x1 - x0; fs0 = s0; fs1 = s1; }
    m0_mine = (int)o[7];
    o[4] |= o[5] << 8;  // (the object's quads, counted above)
  }
  int wsum = fw, ylo = fs1 > fs0 ? fs0 / sft::kGridY : (1 << 20), yhi = fs1 > fs0 ? (fs1 - 1) / sft::kGridY + 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    wsum += __shfl_xor(wsum, d);
    ylo = min(ylo, __shfl_xor(ylo, d));
    yhi = max(yhi, __shfl_xor(yhi, d));
  }
  // (a call whose objects are wider in all than the accumulators hold pixels cannot be drawn in windows of whole rows: the
  //  callers' chunks -- five wireframes, twelve arcs, one circle on a surface of at most 251 pixels -- stay far below)
  if (wsum > kAccPixels) return;
  const int hwin = max(1, min(32, kAccPixels / max(wsum, 1)));
  for (int w0 = ylo; w0 < yhi; w0 += hwin) {
  int bx0 = 0, by0 = 0, bw = 0, bh = 0, S0 = 0, nsub = 0;
  if (lane < nobj) {
    const int s0 = max(fs0, w0 * sft::kGridY), s1 = min(fs1, (w0 + hwin) * sft::kGridY);
    if (fw > 0 && s1 > s0) {
      bx0 = fx0; bw = fw;
      by0 = s0 / sft::kGridY; bh = (s1 - 1) / sft::kGridY + 1 - by0;
      S0 = s0; nsub = s1 - s0;
    }
  }
  int abase = bw * bh, rbase = bh, sbase = nsub;  // inclusive scans
#pragma unroll
  for (int d = 1; d < kMaxObjs; d <<= 1) {  // (lanes 0 .. kMaxObjs - 1 hold the objects; the rest carry zeros)
    const int a = __shfl_up(abase, d), r = __shfl_up(rbase, d), s = __shfl_up(sbase, d);
    if (lane >= d) { abase += a; rbase += r; sbase += s; }
  }
  const int tot_rows = __builtin_amdgcn_readlane(rbase, kMaxObjs - 1), tot_sub = __builtin_amdgcn_readlane(sbase, kMaxObjs - 1);
  const int tot_pix = __builtin_amdgcn_readlane(abase, kMaxObjs - 1);
  if (lane < nobj) {
    uint32_t* o = C.obj(lane);
    o[0] = (uint32_t)bx0 | ((uint32_t)by0 << 8) | ((uint32_t)bw << 16) | ((uint32_t)bh << 24);
    o[1] = (uint32_t)S0 | ((uint32_t)m0_mine << 16);    // (+ a circle's first half in the upper bits)
    o[2] = (uint32_t)nsub;
    o[3] = (uint32_t)(abase - bw * bh);         // where the object's accumulator starts (pixels)
    o[5] = (uint32_t)(rbase - bh);             // where its rows start in the enumeration of rows
    o[6] = 0u;                                  // its rows' modes (bit r: row by0 + r is taken whole)
    o[7] = (uint32_t)(sbase - nsub);           // where its sub-rows start
  }
  C.sync();
  if (SFTD_STOP == 1) return;
  // (the scans as per-lane constants for the searches below: the start of object j's rows / sub-rows / pixels)
  auto find_obj = [&](int t, int which) {  // the object whose range of the enumeration holds t
    int o = 0;
    for (int j = 1; j < nobj; j++) o += t >= (int)C.obj(j)[which] ? 1 : 0;
    return o;
  };
  // ---- rows: the mode of each, and the rows taken whole
  for (int base = 0; base < tot_rows; base += 64) {
    const int t = base + lane;
    if (t >= tot_rows) continue;
    const int o = find_obj(t, 5);
    const uint32_t* ob = C.obj(o);
    const int obx0 = (int)(ob[0] & 255u), oby0 = (int)((ob[0] >> 8) & 255u), obw = (int)((ob[0] >> 16) & 255u);
    const int q0 = (int)(ob[4] & 255u), nq = (int)((ob[4] >> 8) & 255u), okind = (int)((ob[4] >> 16) & 255u);
    const int r = t - (int)ob[5], row = oby0 + r, s0 = row * sft::kGridY, m0 = (int)(ob[1] >> 16);
    bool full = true;
    // every quad's two edges through the row: cells at the row's top, one sub-row above, at the next row's top; order key
    int n_act = 0;
    int e_top[2 * MAXACT], e_bot[2 * MAXACT], e_tie[2 * MAXACT], e_new[2 * MAXACT], e_rank[2 * MAXACT];
    int a_quad[MAXACT], a_le[MAXACT], a_re[MAXACT], a_lt[MAXACT], a_lb[MAXACT], a_rt[MAXACT], a_rb[MAXACT];
    int nact_q = 0;
    for (int k = 0; k < nq && full; k++) {
      const RecView rv{C.rec(q0 + k)};
      int gy[4];
#pragma unroll
      for (int v = 0; v < 4; v++) {
        gy[v] = sft::to_grid_y(rv.vy(v));
        full &= !(gy[v] > s0 && gy[v] < s0 + sft::kGridY);
      }
      sft::QuadScan qo;
      const bool ho = rv.has_out();
      if (ho) {
        sft::Quad q;
#pragma unroll
        for (int v = 0; v < 4; v++) { q.x[v] = rv.vx(v); q.y[v] = rv.vy(v); }
        qo = sft::quad_scan(q, XM);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool shared_face = (okind == kKindSingle || okind == kKindRing) && ((e == 1 && k < nq - 1) || (e == 3 && k > 0));
          if (qo.out_s1[e] > qo.out_s0[e] && !shared_face)
            full &= !(qo.out_s0[e] > s0 && qo.out_s0[e] < s0 + sft::kGridY) && !(qo.out_s1[e] > s0 && qo.out_s1[e] < s0 + sft::kGridY);
        }
      }
      if (!full) break;
      if (!(rv.s0() <= s0 && rv.s1() >= s0 + sft::kGridY)) continue;
      int le = -1, re = -1;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if ((rv.horiz() >> e) & 1u) continue;
        const int g0 = min(gy[e], gy[(e + 1) & 3]), g1 = max(gy[e], gy[(e + 1) & 3]);
        if (!(g0 <= s0 && g1 >= s0 + sft::kGridY)) continue;
        if ((rv.left() >> e) & 1u) le = e; else re = e;
      }
      if (le < 0 || re < 0) continue;
      if (nact_q == MAXACT) { full = false; break; }
      auto cells = [&](int e, int* tp, int* bt) {
        const int kRank = e == 0 ? 0 : (e == 1 ? 1 : (e == 2 ? 3 : 2));
        const bool out = ho && s0 >= qo.out_s0[e] && s0 < qo.out_s1[e];
        const int g0 = max(min(gy[e], gy[(e + 1) & 3]), 0);
        int start = g0;
        if (ho) {
          if (out) start = max(qo.out_s0[e], 0);
          else if (qo.out_s1[e] > qo.out_s0[e] && qo.out_s1[e] <= s0 && qo.out_s1[e] > g0) start = qo.out_s1[e];
        }
        int tie;
        if (out) { *tp = *bt = qo.out_x[e]; tie = qo.out_x[e]; }
        else {
          const sft::EdgeAB ab = rv.edge(e);
          *tp = sft::edge_cell(ab, s0);
          *bt = sft::edge_cell(ab, s0 + sft::kGridY);
          tie = sft::edge_cell(ab, s0 - 1);
        }
        const int rank = 8 * k + kRank + (out ? 4 : 0), is_new = start == s0 ? 1 : 0;  // (k < 32: below 256)
        // (the faces between the pieces of a flattened curve bound the pieces but are no edges of cairo's polygon)
        const bool shared_face = (okind == kKindSingle || okind == kKindRing) && ((e == 1 && k < nq - 1) || (e == 3 && k > 0));
        if (!shared_face) {
          e_top[n_act] = *tp; e_bot[n_act] = *bt; e_new[n_act] = is_new; e_rank[n_act] = rank; e_tie[n_act] = is_new ? rank : tie;
          n_act++;
        }
      };
      int lt, lb, rt, rb;
      cells(le, &lt, &lb);
      cells(re, &rt, &rb);
      a_quad[nact_q] = k; a_le[nact_q] = le; a_re[nact_q] = re; a_lt[nact_q] = lt; a_lb[nact_q] = lb; a_rt[nact_q] = rt; a_rb[nact_q] = rb;
      nact_q++;
    }
    if (full) {  // the list's order at the row's top against the cells at the next row's top: every pair
      for (int i = 0; i < n_act && full; i++)
        for (int j = i + 1; j < n_act && full; j++) {
          bool i_first;
          if (e_top[i] != e_top[j]) i_first = e_top[i] < e_top[j];
          else if (e_new[i] != e_new[j]) i_first = e_new[i] < e_new[j];
          else if (e_tie[i] != e_tie[j]) i_first = e_tie[i] < e_tie[j];
          else i_first = e_rank[i] < e_rank[j];
          full &= i_first ? e_bot[i] <= e_bot[j] : e_bot[j] <= e_bot[i];
        }
    }
    if (full && nact_q > 0) {
      __hip_atomic_fetch_or(const_cast<uint32_t*>(ob) + 6, 1u << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // every source with all its members in the row: left = the member edge last in the list, right = the first
      const unsigned multi = multi_sources(okind);
      const int arow = (int)ob[3] + r * obw;
      for (int si = 0; si < nact_q + 4; si++) {  // every quad in the row alone, then the object's overlap sets
        unsigned members = 0u;
        int sign = 1;
        if (si >= nact_q) {
          members = (multi >> (4 * (si - nact_q))) & 15u;
          if (!members) continue;
          sign = (__popc(members) & 1) ? 1 : -1;
        }
        int lq = -1, rq = -1, lt = 0, lb = 0, rt = 0, rb = 0, le = 0, re = 0, cnt = 0;
        for (int a = 0; a < nact_q; a++) {
          if (si < nact_q) { if (a != si) continue; }
          else {
            const int sl = slot_of(okind, a_quad[a], nq, m0);
            if (sl < 0 || !((members >> sl) & 1u)) continue;
          }
          cnt++;
          if (lq < 0 || a_lt[a] > lt || (a_lt[a] == lt && a_lb[a] > lb)) { lq = a_quad[a]; lt = a_lt[a]; lb = a_lb[a]; le = a_le[a]; }
          if (rq < 0 || a_rt[a] < rt || (a_rt[a] == rt && a_rb[a] < rb)) { rq = a_quad[a]; rt = a_rt[a]; rb = a_rb[a]; re = a_re[a]; }
        }
        if (cnt != (si < nact_q ? 1 : __popc(members)) || lt > rt) continue;
        auto edge_of = [&](int q, int e) {
          const RecView rv{C.rec(q0 + q)};
          if (rv.has_out()) {
            sft::Quad qq;
#pragma unroll
            for (int v = 0; v < 4; v++) { qq.x[v] = rv.vx(v); qq.y[v] = rv.vy(v); }
            const sft::QuadScan qo = sft::quad_scan(qq, XM);
            if (s0 >= qo.out_s0[e] && s0 < qo.out_s1[e]) return sft::row_edge(qo.out_x[e], 0, qo.out_x[e], 256, s0);
          }
          return sft::row_edge(rv.vx(e), rv.vy(e), rv.vx((e + 1) & 3), rv.vy((e + 1) & 3), s0);
        };
        const sft::RowEdge EL = edge_of(lq, le), ER = edge_of(rq, re);
        const int c0 = max(EL.ix1, obx0), c1 = min(ER.ix2, obx0 + obw - 1);
        for (int c = c0; c <= c1; c++) {
          const int v = sft::row_edge_area(EL, c) - sft::row_edge_area(ER, c);
          if (v) acc_add(acc, arow + c - obx0, sign * v);
        }
      }
    }
  }
  C.sync();
  if (SFTD_STOP == 2) return;
  // ---- sub-rows of the other rows
  for (int base = 0; base < tot_sub; base += 64) {
    const int t = base + lane;
    if (t >= tot_sub) continue;
    const int o = find_obj(t, 7);
    const uint32_t* ob = C.obj(o);
    const int obx0 = (int)(ob[0] & 255u), oby0 = (int)((ob[0] >> 8) & 255u), obw = (int)((ob[0] >> 16) & 255u);
    const int q0 = (int)(ob[4] & 255u), nq = (int)((ob[4] >> 8) & 255u), okind = (int)((ob[4] >> 16) & 255u);
    const int s = (int)(ob[1] & 0xffffu) + t - (int)ob[7], m0 = (int)(ob[1] >> 16);
    const int row = s / sft::kGridY, r = row - oby0;
    if ((ob[6] >> r) & 1u) continue;  // taken whole
    const int arow = (int)ob[3] + r * obw;
    int sl[4] = {0, 0, 0, 0}, sr[4] = {0, 0, 0, 0};  // the spans kept for the overlaps (slots 0..3)
    for (int k = 0; k < nq; k++) {
      const RecView rv{C.rec(q0 + k)};
      int L = 0, R = 0;
      if (s >= rv.s0() && s < rv.s1()) rec_interval(rv, s, XM, &L, &R);
      add_span(acc, arow, obx0, obw, L, R, 1);
      const int slot = slot_of(okind, k, nq, m0);
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (slot == j) { sl[j] = L; sr[j] = R; }
    }
    const unsigned multi = multi_sources(okind);
#pragma unroll
    for (int si = 0; si < 4; si++) {
      const unsigned members = (multi >> (4 * si)) & 15u;
      if (!members) continue;
      int L = sft::kCellMin, R = sft::kCellMax;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((members >> j) & 1u) {
          // (an inactive or empty member has R <= L: the intersection is empty)
          if (sr[j] <= sl[j]) { L = 1; R = 0; }
          L = max(L, sl[j]);
          R = min(R, sr[j]);
        }
      add_span(acc, arow, obx0, obw, L, R, (__popc(members) & 1) ? 1 : -1);
    }
  }
  C.sync();
  if (SFTD_STOP == 3) return;
  // ---- pixels: object after object (the reference composites its strokes in order; boxes of different objects may overlap)
  for (int o = 0; o < nobj; o++) {
    const uint32_t* ob = C.obj(o);
    const int obx0 = (int)(ob[0] & 255u), oby0 = (int)((ob[0] >> 8) & 255u), obw = (int)((ob[0] >> 16) & 255u), obh = (int)(ob[0] >> 24);
    const int ab = (int)ob[3], ogrey = (int)(ob[4] >> 24), n = obw * obh;
    for (int i = lane; i < n; i += 64) {
      const int ry = i / obw, rx = i - ry * obw;
      const int cov = (int)((acc[(ab + i) >> 1] >> (16 * ((ab + i) & 1))) & 0xffffu);
      const int a = sft::area_to_alpha(cov);
      if (a) {
        uint8_t* p = C.fb + (oby0 + ry) * C.W + obx0 + rx;
        *p = (uint8_t)sft::lerp8(ogrey, a, *p);
      }
    }
    C.sync();
  }
  for (int i = lane; i < (tot_pix + 1) / 2; i += 64) acc[i] = 0u;
  C.sync();
  }  // (the next window of rows)
}


// =====================================================================================================================================
// The FAST arrangement (the default geometry's frame kernel): the same arithmetic, organised for the common case -- objects that
// stay inside the surface's left and right borders, quads with at most two edges on either side (every stroke rectangle, every
// arc piece).  A call that holds anything else goes through raster() above (the caller asks needs_general()).
//   * a quad's record holds its edges SORTED: the two left ones, then the two right ones (a side with one edge holds it twice), as
//     (A, B) with the rounding folded in: cell = low word of (A + B s + 1.5 * 2^52) -- two float64 operations per edge;
//   * the rows that contain a vertex are known from a bit map the quads' owners OR together (no per-row search);
//   * a row taken whole is handed to one lane per (row, quad) for its two trapezoid edges; rows in which two quads of an object
//     overlap while taken whole (rare: the overlap of two lines is shorter than a pixel) go to one lane with the general code;
//   * spans are added with straight-line code for the one- and two-pixel cases.
#ifndef SFTD_FAST_QUADS
#define SFTD_FAST_QUADS 16 /* quads per call of the fast arrangement (A/B: 32 -- fewer calls in crowded frames, 4.4 KB more LDS) */
#endif
constexpr int kMaxQuadsF = SFTD_FAST_QUADS;
constexpr int kAccPixelsF = SFTD_FAST_QUADS > 16 ? 1024 : kAccPixels;  // (ten wireframes' boxes instead of five)
constexpr int kRecWordsF = 48;
// A quad's record:
//    0 .. 15   the slots L1 L2 R1 R2: an edge as two doubles (A - 1/2, B): cell(s) = round(A' + B s)   (cell_fast)
//   16 17      the quad's sub-rows [s0, s1)
//   18         the slots' edge numbers (2 bits each, slots 0 .. 5) | left edges << 12 | right edges << 14 | kRecBorder | kRecThird
//   19         the rows slots L3 | R3 << 16 span whole (as 46, 47)
//   20 .. 31   gy[4], x[4], y[4]: the corners
//   32 .. 39   the slots L3 R3: a third edge on one side (the curve pieces of an explosion; else a copy of L1 / R1)
//   40 .. 45   a slot's piece along the surface's border (cairo-polygon.c: _add_clipped_edge): in the sub-rows [b0, b1) the edge is
//              the vertical x = 0 / x = the surface's width: b0 | b1 << 12 | right border << 24; 0 = none
//   46 47      the pixel rows a slot's edge spans whole, [lo, hi) as lo | hi << 8: L1 | L2 << 16, R1 | R2 << 16
constexpr unsigned kRecBorder = 1u << 16, kRecThird = 1u << 17;
constexpr int kHdrWordsF = 8;   // per RUN of sub-rows (up to two per quad: above and below its rows taken whole), for the sub-rows' lanes; word 5: see raster_fast
constexpr int kObjWordsF = 12;  // as kObjWords, + the rows that hold a vertex: 96 bits
constexpr int kTasksF = 64;     // (row, quad) pairs taken whole, per round
constexpr int kMapWordsF = 64;  // the sub-rows' enumeration: a bit per start of a quad's run, 2 048 sub-rows per round
static_assert(kMapWordsF <= kTasksF + 8, "the sub-rows' map lies over the rows' task list (one is dead when the other is written)");
constexpr int kAccAtF = kMaxQuadsF * (kRecWordsF + 2 * kHdrWordsF) + kMaxObjs * kObjWordsF + kTasksF + 8;
constexpr int kLdsWordsF = kAccAtF + kAccPixelsF / 2;

struct CtxF {
  uint32_t* lds;
  uint8_t* fb;
  int W, H, lane;
  __device__ __forceinline__ uint32_t* rec(int q) const { return lds + q * kRecWordsF; }
  __device__ __forceinline__ uint32_t* hdr(int q) const { return lds + kMaxQuadsF * kRecWordsF + q * kHdrWordsF; }
  __device__ __forceinline__ uint32_t* obj(int o) const { return lds + kMaxQuadsF * (kRecWordsF + 2 * kHdrWordsF) + o * kObjWordsF; }
  __device__ __forceinline__ uint32_t* tasks() const { return obj(kMaxObjs); }  // [kTasksF], the count, "more rows than fit", (2 free)
  __device__ __forceinline__ uint32_t* map() const { return tasks(); }  // (the rows' pass is over when the sub-rows' map is made)
  __device__ __forceinline__ uint32_t* acc() const { return lds + kAccAtF; }
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};

// inclusive prefix sum across the wave: six DPP adds (row_shr 1, 2, 4, 8; row_bcast 15 into rows 1 and 3, 31 into rows 2 and 3) --
// no trips through LDS (__shfl_up is a ds_bpermute a step: six dependent LDS round trips for one scan)
#ifndef SFTD_DPP_SCANS
#define SFTD_DPP_SCANS 1
#endif
__device__ __forceinline__ int wave_incl_sum(int v) {
#if SFTD_DPP_SCANS
#define SFTD_SCAN_STEP(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, (ctrl), (rmask), 0xf, false)
  SFTD_SCAN_STEP(0x111, 0xf);
  SFTD_SCAN_STEP(0x112, 0xf);
  SFTD_SCAN_STEP(0x114, 0xf);
  SFTD_SCAN_STEP(0x118, 0xf);
  SFTD_SCAN_STEP(0x142, 0xa);
  SFTD_SCAN_STEP(0x143, 0xc);
#undef SFTD_SCAN_STEP
  return v;
#else
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int a = __shfl_up(v, d);
    if (lane >= d) v += a;
  }
  return v;
#endif
}

constexpr double kMagic52 = 6755399441055744.0;  // 1.5 * 2^52: adding it leaves round-to-nearest(x) in the low word
__device__ __forceinline__ int cell_fast(const uint32_t* slot, double sd) {  // slot = (A - 1/2, B) as two doubles; sd = (double)s
  const double2 ab = *reinterpret_cast<const double2*>(slot);
  return __double2loint(fma(ab.y, sd, ab.x) + kMagic52);
}
__device__ __forceinline__ int slot_at(int slot) { return slot < 4 ? 4 * slot : 16 + 4 * slot; }  // (slots 4, 5 at words 32, 36)
// a slot's cell where its border piece stands in for it
__device__ __forceinline__ int border_cell(int c, unsigned bw, int s, int xmax) {
  const int b0 = (int)(bw & 4095u), b1 = (int)((bw >> 12) & 4095u);
  return (s >= b0 && s < b1) ? ((bw >> 24) ? xmax : 0) : c;
}
// the quad's span [L, R) in sub-row s (sd = (double)s); `on`: this lane asks (the rare parts sit behind a ballot)
__device__ __forceinline__ void span_fast(const uint32_t* rc, int s, double sd, bool on, int xmax, int* L, int* R) {
  int l1 = cell_fast(rc, sd), l2 = cell_fast(rc + 4, sd), r1 = cell_fast(rc + 8, sd), r2 = cell_fast(rc + 12, sd);
  const bool ext = on && (rc[18] & (kRecBorder | kRecThird));
  if (__ballot(ext)) {
    if (ext) {
      int l3 = cell_fast(rc + 32, sd), r3 = cell_fast(rc + 36, sd);
      const uint4 b = *reinterpret_cast<const uint4*>(rc + 40);
      const uint2 b3 = *reinterpret_cast<const uint2*>(rc + 44);
      l1 = border_cell(l1, b.x, s, xmax); l2 = border_cell(l2, b.y, s, xmax);
      r1 = border_cell(r1, b.z, s, xmax); r2 = border_cell(r2, b.w, s, xmax);
      l3 = border_cell(l3, b3.x, s, xmax); r3 = border_cell(r3, b3.y, s, xmax);
      l1 = max(l1, l3); r1 = min(r1, r3);
    }
  }
  *L = max(l1, l2);
  *R = min(r1, r2);
}

__device__ __forceinline__ void add_span_fast(uint32_t* acc, int arow, int bx0, int bw, int L, int R, int sign) {
  if (R <= L) return;
  const int p0 = max(L >> 8, bx0), p1 = min((R - 1) >> 8, bx0 + bw - 1);
  if (p0 > p1) return;
  const int lo = max(L, p0 << 8);
  if (p0 == p1) {
    acc_add(acc, arow + p0 - bx0, sign * 2 * (min(R, (p0 + 1) << 8) - lo));
    return;
  }
  acc_add(acc, arow + p0 - bx0, sign * 2 * (((p0 + 1) << 8) - lo));
  acc_add(acc, arow + p1 - bx0, sign * 2 * (min(R, (p1 + 1) << 8) - (p1 << 8)));
  for (int px = p0 + 1; px < p1; px++) acc_add(acc, arow + px - bx0, sign * 512);
}

// a slot's edge through the row that starts at sub-row s0, taken whole, as cairo's trapezoid edge
__device__ __forceinline__ sft::RowEdge rec_row_edge(const uint32_t* rc, int slot, int s0, int xmax) {
  const unsigned bw = rc[40 + slot];
  if (s0 >= (int)(bw & 4095u) && s0 < (int)((bw >> 12) & 4095u)) {  // along the border: a vertical
    const int bx = (bw >> 24) ? xmax : 0;
    return sft::row_edge_ab(sft::EdgeAB{0.0, 0.0}, 0, 256, bx, s0);
  }
  const int e = (int)((rc[18] >> (2 * slot)) & 3u), j = (e + 1) & 3;
  int x1 = (int)rc[24 + e], y1 = (int)rc[28 + e], x2 = (int)rc[24 + j], y2 = (int)rc[28 + j];
  if (y2 < y1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
  const double2 ab = *reinterpret_cast<const double2*>(rc + slot_at(slot));
  return sft::row_edge_ab(sft::EdgeAB{ab.x + 0.5, ab.y}, x2 - x1, y2 - y1, x1, s0);
}
// the slots (left, right) whose edges span pixel row `row` whole, or -1
__device__ __forceinline__ void row_slots(const uint32_t* rc, int row, int* ls, int* rs) {
  const uint2 rw = *reinterpret_cast<const uint2*>(rc + 46);
  const unsigned r3 = rc[19];
  auto in = [&](unsigned h) { return row >= (int)(h & 255u) && row < (int)((h >> 8) & 255u); };
  *ls = in(rw.x) ? 0 : (in(rw.x >> 16) ? 1 : (in(r3) ? 4 : -1));
  *rs = in(rw.y) ? 2 : (in(rw.y >> 16) ? 3 : (in(r3 >> 16) ? 5 : -1));
}

// `mine`: this lane's quad (valid lanes); obj0: the first lane of its object (lanes of an object are consecutive)
#ifndef SFTD_FAST_CALL
#define SFTD_FAST_CALL 0 /* A/B: 1 = raster_fast as ONE function the frame kernel calls from its twelve places instead of twelve copies */
#endif
// what the pixels' pass needs of a call whose earlier passes ran apart from it (raster_fast<1>, then raster_fast_pixels: the frame
// kernel's explosion pre-pass has eight waves rasterise a ring each into accumulators of their own and composite in draw order)
struct RasterCarry {
  int nobj, tot_pix;
};
__device__ __forceinline__ void raster_fast_pixels(const CtxF& C, int nobj, int tot_pix);
#if SFTD_FAST_CALL
__device__ __attribute__((noinline)) void raster_fast(const CtxF C, const sft::Quad mine, bool valid, int obj0, int kind, int grey) {
  constexpr int PHASE = 0;
  RasterCarry* const carry = nullptr;
#else
template <int PHASE = 0>  // 0: the whole call; 1: everything but the pixels' pass (*carry says what is left)
__device__ __forceinline__ void raster_fast(const CtxF& C, const sft::Quad& mine, bool valid, int obj0, int kind, int grey,
                                            RasterCarry* carry = nullptr) {
#endif
  if (PHASE == 1) *carry = RasterCarry{0, 0};
  if (SFTD_STOP == 0) return;
  const int lane = C.lane, xmax = C.W * 256;
  uint32_t* const acc = C.acc();
  const unsigned long long vmask = __ballot(valid);
  if (!vmask) return;
  const int qi = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vmask, 0u));
  const bool leader = valid && lane == obj0;
  const unsigned long long lmask = __ballot(leader);
  const int oi = (int)__popcll(lmask & ((2ull << obj0) - 1ull)) - 1;
  const int nobj = (int)__popcll(lmask);
  // (this quad's place in its object: number, and whether it is the last)
  const unsigned long long from_obj0 = ~((1ull << obj0) - 1ull);
  const unsigned long long later_leaders = lmask & ~((2ull << obj0) - 1ull);
  const unsigned long long obj_lanes = vmask & from_obj0 & (later_leaders ? ((later_leaders & (~later_leaders + 1ull)) - 1ull) : ~0ull);
  const int kq = (int)__popcll(obj_lanes & ((1ull << lane) - 1ull)), nq_mine = (int)__popcll(obj_lanes);
  // ---- records
  if (leader) {
    uint32_t* o = C.obj(oi);
    o[0] = 0x7fffffffu; o[1] = 0x7fffffffu; o[2] = 0x80000000u; o[3] = 0x80000000u;  // min x, min s, max x, max s
    o[4] = (uint32_t)qi | ((uint32_t)nq_mine << 8) | ((uint32_t)(kind & 255) << 16) | ((uint32_t)grey << 24);
    o[5] = 0u; o[6] = 0u; o[7] = (uint32_t)(kind >> 8);
    o[8] = 0u; o[9] = 0u; o[10] = 0u; o[11] = 0u;
  }
  C.sync();
  int my_lo = 0, my_hi = 0;
  if (valid) {
    uint32_t* r = C.rec(qi);
    int gy[4], lo = 1 << 30, hi = -(1 << 30), minx = 1 << 30, maxx = -(1 << 30);
    unsigned vr0 = 0u, vr1 = 0u, vr2 = 0u;
    // the row that holds sub-row boundary g strictly inside (a vertex on a row's boundary is no event for either row)
    auto event_row = [&](int g) {
      const int vrow = (g * 34953) >> 19;  // g / 15 for 0 <= g < 2^16
      if (g > 0 && vrow < 96 && g != vrow * sft::kGridY) {
        vr0 |= vrow < 32 ? 1u << vrow : 0u;
        vr1 |= (vrow >= 32 && vrow < 64) ? 1u << (vrow - 32) : 0u;
        vr2 |= vrow >= 64 ? 1u << (vrow - 64) : 0u;
      }
    };
#pragma unroll
    for (int k = 0; k < 4; k++) {
      gy[k] = sft::to_grid_y(mine.y[k]);
      lo = min(lo, gy[k]); hi = max(hi, gy[k]);
      minx = min(minx, mine.x[k]); maxx = max(maxx, mine.x[k]);
      event_row(gy[k]);
    }
    *reinterpret_cast<int4*>(r + 20) = int4{gy[0], gy[1], gy[2], gy[3]};
    *reinterpret_cast<int4*>(r + 24) = int4{mine.x[0], mine.x[1], mine.x[2], mine.x[3]};
    *reinterpret_cast<int4*>(r + 28) = int4{mine.y[0], mine.y[1], mine.y[2], mine.y[3]};
    *reinterpret_cast<uint4*>(r + 40) = uint4{0u, 0u, 0u, 0u};
    *reinterpret_cast<uint4*>(r + 44) = uint4{0u, 0u, 0u, 0u};
    r[19] = 0u;
    // which side an edge is on: with the corners in order, the edges that go down are all on one side (the right one when the
    // doubled area (x2 - x0)(y3 - y1) - (x3 - x1)(y2 - y0) is positive: y points down)
    const int area2 = (mine.x[2] - mine.x[0]) * (mine.y[3] - mine.y[1]) - (mine.x[3] - mine.x[1]) * (mine.y[2] - mine.y[0]);
    // a parallelogram's opposite edges share their slope: two reciprocals instead of four
    const bool para = mine.x[1] - mine.x[0] == mine.x[2] - mine.x[3] && mine.y[1] - mine.y[0] == mine.y[2] - mine.y[3];
    double Kc[4], Qc[4];  // K = dx / (30 dy) of the line (the same whichever way the edge is read); Q = 1 / (8 * 30 |dy|)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j = (k + 1) & 3;
      const int dx = mine.x[j] - mine.x[k], dy = mine.y[j] - mine.y[k];
      if (k >= 2 && para) { Kc[k] = Kc[k - 2]; Qc[k] = Qc[k - 2]; }
      else {
        const double rd = dy != 0 ? sft::rcp(30.0 * (double)dy) : 0.0;
        Kc[k] = (double)dx * rd;
        Qc[k] = 0.125 * fabs(rd);
      }
    }
    const bool chain = (kind & 255) == kKindSingle || (kind & 255) == kKindRing;
    int nl = 0, nr = 0;
    unsigned flags = 0u;
    uint16_t* const rows_lo = reinterpret_cast<uint16_t*>(r + 46);  // [4]: slots 0 .. 3
    uint16_t* const rows_hi = reinterpret_cast<uint16_t*>(r + 19);  // [2]: slots 4, 5
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j = (k + 1) & 3;
      const int dy = mine.y[j] - mine.y[k];
      if (dy == 0 || area2 == 0) continue;
      const bool right = (dy > 0) == (area2 > 0);
      const int n = right ? nr : nl;
      const int slot = right ? (n == 0 ? 2 : (n == 1 ? 3 : 5)) : (n == 0 ? 0 : (n == 1 ? 1 : 4));
      if (right) nr++; else nl++;
      const bool down = dy > 0;
      const int x1 = down ? mine.x[k] : mine.x[j], y1 = down ? mine.y[k] : mine.y[j];  // the upper end
      const int x2 = down ? mine.x[j] : mine.x[k], y2 = down ? mine.y[j] : mine.y[k];
      const int g1 = down ? gy[k] : gy[j], g2 = down ? gy[j] : gy[k];
      // cell(s) = round(A + B s): A = x + (256 - 30 y) K + a quarter of the least distance to a rounding boundary (sf_tor.h: edge_ab)
      const double A = (double)x1 + (256.0 - 30.0 * (double)y1) * Kc[k] + Qc[k];
      *reinterpret_cast<double2*>(r + slot_at(slot)) = double2{A, 512.0 * Kc[k]};
      flags |= (unsigned)k << (2 * slot);
      // the pixel rows the edge spans whole
      const int r_lo = g1 <= 0 ? 0 : ((g1 + 14) * 34953) >> 19, r_hi = g2 <= 0 ? 0 : min((g2 * 34953) >> 19, 255);
      const uint16_t rows = (uint16_t)(r_lo | (max(r_hi, r_lo) << 8));
      if (slot < 4) rows_lo[slot] = rows; else rows_hi[slot - 4] = rows;
    }
    // where an edge leaves the surface sideways the polygon runs along the border (sf_tor.h: quad_scan) -- rare: a loop over
    // the slots, one copy of the code.  _cairo_fixed_mul_div_floor's C division as a float64 one: numerator below 2^31,
    // denominator below 2^16 -- a quotient that is no whole number is 2^-16 or more off one.
    if (minx < 0 || maxx > xmax || minx >= xmax || maxx <= 0) {
      auto muldiv = [](int p, int q, int d) { return (int)trunc((double)p * (double)q / (double)d); };
#pragma unroll 1
      for (int slot = 0; slot < 6; slot++) {
        const int n = slot == 0 ? 0 : (slot == 1 ? 1 : (slot == 4 ? 2 : (slot == 2 ? 0 : (slot == 3 ? 1 : 2))));
        if (n >= ((slot == 0 || slot == 1 || slot == 4) ? nl : nr)) continue;
        const int k = (int)((flags >> (2 * slot)) & 3u), j = (k + 1) & 3;
        const bool down = (int)r[28 + j] > (int)r[28 + k];
        const int x1 = (int)r[24 + (down ? k : j)], y1 = (int)r[28 + (down ? k : j)], g1 = (int)r[20 + (down ? k : j)];  // the upper end
        const int x2 = (int)r[24 + (down ? j : k)], y2 = (int)r[28 + (down ? j : k)], g2 = (int)r[20 + (down ? j : k)];
        const int pl = min(x1, x2), pr = max(x1, x2);
        if (!(pr <= 0 || pl >= xmax || pl < 0 || pr > xmax)) continue;
        int b0 = 0, b1 = 0, bx = 0;
        const bool down_right = x1 <= x2;
        if (pr <= 0 || pl >= xmax) { bx = pr <= 0 ? 0 : 1; b0 = g1; b1 = g2; }
        else {
          const int xb = pl < 0 ? 0 : xmax;
          bx = pl < 0 ? 0 : 1;
          // _cairo_edge_compute_intersection_y_for_x, then x_for_y of that y: one more step if it is still outside
          int y = xb == x1 ? y1 : (xb == x2 ? y2 : y1 + muldiv(xb - x1, y2 - y1, x2 - x1));
          const int xy = y == y1 ? x1 : (y == y2 ? x2 : x1 + muldiv(y - y1, x2 - x1, y2 - y1));
          if (bx == 0 ? xy < 0 : xy > xmax) y += (down_right == (bx == 0)) ? 1 : -1;
          y = y < y1 ? y1 : (y > y2 ? y2 : y);
          const int gc = sft::to_grid_y(y);
          // (outside above the crossing: left border with x growing downwards, right border with x shrinking downwards)
          if (down_right == (bx == 0)) { b0 = g1; b1 = gc; } else { b0 = gc; b1 = g2; }
        }
        if (b1 > b0) {
          // (the end of the piece inside the edge is one more vertex of the polygon -- unless the edge is a face between two
          //  pieces of a flattened curve: no edge of the polygon, nothing is clipped)
          if (!(chain && ((k == 1 && kq < nq_mine - 1) || (k == 3 && kq > 0)))) { event_row(b0); event_row(b1); }
          r[40 + slot] = (uint32_t)min(max(b0, 0), 4095) | ((uint32_t)min(max(b1, 0), 4095) << 12) | ((uint32_t)bx << 24);
          flags |= kRecBorder;
        }
      }
    }
    // a side with one edge holds it twice (max / min of a value with itself) -- the copies span no rows --, the third slots hold
    // copies unless a side has three edges; a degenerate quad (no edge at all on a side) spans nothing
    if (nl == 1) { *reinterpret_cast<double2*>(r + 4) = *reinterpret_cast<const double2*>(r); r[41] = r[40]; }
    if (nr == 1) { *reinterpret_cast<double2*>(r + 12) = *reinterpret_cast<const double2*>(r + 8); r[43] = r[42]; }
    if (nl < 3) { *reinterpret_cast<double2*>(r + 32) = *reinterpret_cast<const double2*>(r); r[44] = r[40]; }
    if (nr < 3) { *reinterpret_cast<double2*>(r + 36) = *reinterpret_cast<const double2*>(r + 8); r[45] = r[42]; }
    if (nl == 3 || nr == 3) flags |= kRecThird;
    if (nl == 0 || nr == 0) hi = lo;
    my_lo = lo; my_hi = hi;
    r[16] = (uint32_t)lo; r[17] = (uint32_t)hi; r[18] = flags | ((unsigned)nl << 12) | ((unsigned)nr << 14);
    int* o = reinterpret_cast<int*>(C.obj(oi));
    __hip_atomic_fetch_min(o + 0, minx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_min(o + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(o + 2, maxx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(o + 3, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (vr0) __hip_atomic_fetch_or(C.obj(oi) + 8, vr0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (vr1) __hip_atomic_fetch_or(C.obj(oi) + 9, vr1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (vr2) __hip_atomic_fetch_or(C.obj(oi) + 10, vr2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  C.sync();
  // ---- the objects' boxes (lane o < nobj) and the scans over them
  int bx0 = 0, by0 = 0, bw = 0, bh = 0;
  if (lane < nobj) {
    const int* o = reinterpret_cast<const int*>(C.obj(lane));
    const int x0 = max(o[0] >> 8, 0), x1 = min((o[2] + 255) >> 8, C.W);
    const int s0 = max(o[1], 0), s1 = min(o[3], C.H * sft::kGridY);
    if (x1 > x0 && s1 > s0) {
      bx0 = x0; bw = x1 - x0;
      by0 = (s0 * 34953) >> 19; bh = (((s1 - 1) * 34953) >> 19) + 1 - by0;
    }
  }
  // (lanes nobj .. 63 carry zeros: the whole-wave scans give lanes 0 .. kMaxObjs - 1 what a sixteen-lane one would)
  const int abase = wave_incl_sum(bw * bh), rbase = wave_incl_sum(bh);
  const int tot_rows = __builtin_amdgcn_readlane(rbase, kMaxObjs - 1), tot_pix = __builtin_amdgcn_readlane(abase, kMaxObjs - 1);
  if (lane < nobj) {
    uint32_t* o = C.obj(lane);
    o[0] = (uint32_t)bx0 | ((uint32_t)by0 << 8) | ((uint32_t)bw << 16) | ((uint32_t)bh << 24);
    o[1] = o[7] << 16;            // (a circle's first half)
    o[3] = (uint32_t)(abase - bw * bh);
    o[5] = (uint32_t)(rbase - bh);
    o[6] = 0u;
  }
  C.sync();
  if (SFTD_STOP == 1) return;
  // (row t of the enumeration belongs to the last object whose rows start at or before it: the starts come out of lanes
  //  0 .. nobj - 1's registers -- scalar reads, no round trips to LDS)
  const int row_start = rbase - bh;
  auto find_obj = [&](int t, int) {
    int o = 0;
    for (int j = 1; j < nobj; j++) o += t >= __builtin_amdgcn_readlane(row_start, j) ? 1 : 0;
    return o;
  };
  // ---- rows: which are taken whole (can_do_full_row and the buckets: no vertex strictly inside the row, no two edges that change
  // places); their (row, quad) pairs go on the task list, the list is worked off by a lane per pair.  A list that is full (never
  // with this game's objects) leaves its rows for another round.
  for (bool more = true; more;) {
    if (lane == 0) { C.tasks()[kTasksF] = 0u; C.tasks()[kTasksF + 1] = 0u; }
    C.sync();
    for (int base = 0; base < tot_rows; base += 64) {
      const int t = base + lane;
      const bool have = t < tot_rows;
      const int o = have ? find_obj(t, 5) : 0;
      const uint32_t* ob = C.obj(o);
      const int oby0 = (int)((ob[0] >> 8) & 255u), q0 = (int)(ob[4] & 255u), nq = (int)((ob[4] >> 8) & 255u), okind = (int)((ob[4] >> 16) & 255u);
      const int r = t - (int)ob[5], row = oby0 + r, s0 = row * sft::kGridY;
      const unsigned vbits = row < 32 ? ob[8] : (row < 64 ? ob[9] : ob[10]);
      // (rows 32 and up of an object -- none of this game's is that tall -- go by sub-rows; a row of an earlier round is done)
      bool full = have && r < 32 && !((vbits >> (row & 31)) & 1u) && !((ob[6] >> (r & 31)) & 1u);
      int nsp = 0;
      unsigned sp0 = 0u, sp1 = 0u, sp2 = 0u, sp3 = 0u;  // the quads through the row: k | left slot << 8 | right slot << 11
      if (__ballot(full)) {
        for (int k = 0; __ballot(full && k < nq); k++) {
          if (!(full && k < nq)) continue;
          const uint32_t* rc = C.rec(q0 + k);
          const int2 rg = *reinterpret_cast<const int2*>(rc + 16);
          if (!(rg.x <= s0 && rg.y >= s0 + sft::kGridY)) continue;
          int ls, rs;
          row_slots(rc, row, &ls, &rs);
          if (ls < 0 || rs < 0) continue;
          if (nsp == 4) { full = false; continue; }
          const unsigned ent = (unsigned)k | ((unsigned)ls << 8) | ((unsigned)rs << 11);
          sp0 = nsp == 0 ? ent : sp0; sp1 = nsp == 1 ? ent : sp1; sp2 = nsp == 2 ? ent : sp2; sp3 = nsp == 3 ? ent : sp3;
          nsp++;
        }
      }
      // The edges through the row must keep their order, and quads that overlap make the row a "complex" one.
      // An edge's place in cairo's list at the row's top: by cell; equal cells keep the order of one sub-row earlier; an edge that
      // starts with this row comes behind the ones already there, edges that start together in the polygon's edge order: one
      // 64-bit key.  The cells at the next row's top must not decrease in that order -- the two edges of one quad included.
      bool complex_row = false;
      if (__ballot(full && nsp >= 1)) {
        // (how many quads a row of this round has at most: nearly always one or two; three or four: an explosion's circle)
        const int lvl = __ballot(full && nsp > 2) ? 4 : (__ballot(full && nsp > 1) ? 2 : 1);
        if (full && nsp >= 1) {
          const unsigned sp[4] = {sp0, sp1, sp2, sp3};
          const double sd0 = (double)s0, sd1 = (double)(s0 + sft::kGridY), sdm = (double)(s0 - 1);
          unsigned long long key[8];
          int bot[8];
          unsigned in_order = 0u;
#pragma unroll
          for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int side = 0; side < 2; side++) {
              const int i = 2 * a + side;
              key[i] = 0ull; bot[i] = 0;
              if (a < lvl && a < nsp) {
                const int k = (int)(sp[a] & 255u), sl = (int)((sp[a] >> (side ? 11 : 8)) & 7u);
                const uint32_t* rc = C.rec(q0 + k);
                const int e = (int)((rc[18] >> (2 * sl)) & 3u);
                const unsigned bwd = rc[40 + sl];
                const int b0 = (int)(bwd & 4095u), b1 = (int)((bwd >> 12) & 4095u);
                const bool out = s0 >= b0 && s0 < b1;
                const int g0 = max(min((int)rc[20 + e], (int)rc[20 + ((e + 1) & 3)]), 0);
                // the piece of the edge that is in the list: along the border, or the edge proper (from its top / the border piece's end)
                const int begin = out ? b0 : ((b1 > b0 && b1 <= s0 && b1 > g0) ? b1 : g0);
                const int bx = (bwd >> 24) ? xmax : 0;
                const int top = out ? bx : cell_fast(rc + slot_at(sl), sd0);
                bot[i] = out ? bx : cell_fast(rc + slot_at(sl), sd1);
                const int rank = 8 * k + (e == 0 ? 0 : (e == 1 ? 1 : (e == 2 ? 3 : 2))) + (out ? 4 : 0);
                const bool is_new = begin == s0;
                const int tie = is_new ? rank : (out ? bx : cell_fast(rc + slot_at(sl), sdm)) + 65536;
                key[i] = ((unsigned long long)(unsigned)(top + 65536) << 26) | ((unsigned long long)is_new << 25) |
                         ((unsigned long long)(unsigned)tie << 7) | (unsigned long long)rank;
                // (a face between two pieces of a flattened curve is no edge of the polygon: out of the order test)
                if (!((okind == kKindSingle || okind == kKindRing) && ((e == 1 && k < nq - 1) || (e == 3 && k > 0)))) in_order |= 1u << i;
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = i + 1; j < 8; j++) {
              if (j >= 2 * lvl) continue;
              const bool both = ((in_order >> i) & (in_order >> j) & 1u) != 0u;
              const bool ok = key[i] < key[j] ? bot[i] <= bot[j] : bot[j] <= bot[i];
              full = full && (!both || ok);
            }
          // overlapping quads (their spans at the row's top touch or cross): inclusion-exclusion terms needed
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = a + 1; b < 4; b++) {
              if (b >= lvl) continue;
              if (b < nsp) complex_row |= (key[2 * b] >> 26) <= (key[2 * a + 1] >> 26) && (key[2 * a] >> 26) <= (key[2 * b + 1] >> 26);
            }
        }
      }
      if (full) {
        const int n_tasks = complex_row ? 1 : nsp;
        unsigned at = 0u;
        if (n_tasks) at = __hip_atomic_fetch_add(C.tasks() + kTasksF, (unsigned)n_tasks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (at + (unsigned)n_tasks <= (unsigned)kTasksF) {
          __hip_atomic_fetch_or(const_cast<uint32_t*>(ob) + 6, 1u << r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const unsigned head = (unsigned)o | ((unsigned)r << 8);
          const unsigned sp[4] = {sp0, sp1, sp2, sp3};
#pragma unroll
          for (int a = 0; a < 4; a++)
            if (a < n_tasks)
              C.tasks()[at + a] = head | (complex_row ? 0x80000000u : ((sp[a] & 255u) << 16) | (((sp[a] >> 8) & 63u) << 24));
        } else {
          C.tasks()[kTasksF + 1] = 1u;  // another round
        }
      }
    }
    C.sync();
    // ---- the rows taken whole: a lane per (row, quad): the area between the quad's two edges (cell_list_render_edge)
    {
      const unsigned n_tasks = min(C.tasks()[kTasksF], (unsigned)kTasksF);  // (a row that did not fit has added to the count only)
      more = C.tasks()[kTasksF + 1] != 0u;
      const unsigned tk = (unsigned)lane < n_tasks ? C.tasks()[lane] : 0u;
      const bool on = (unsigned)lane < n_tasks;
      const uint32_t* ob = C.obj((int)(tk & 255u));
      const int obx0 = (int)(ob[0] & 255u), oby0 = (int)((ob[0] >> 8) & 255u), obw = (int)((ob[0] >> 16) & 255u);
      const int q0 = (int)(ob[4] & 255u), nq = (int)((ob[4] >> 8) & 255u), okind = (int)((ob[4] >> 16) & 255u), m0 = (int)(ob[1] >> 16);
      const int r = (int)((tk >> 8) & 255u), row = oby0 + r, s0 = row * sft::kGridY, arow = (int)ob[3] + r * obw;
      if (__ballot(on && !(tk >> 31))) {
        if (on && !(tk >> 31)) {
          const uint32_t* rc = C.rec(q0 + (int)((tk >> 16) & 255u));
          const sft::RowEdge EL = rec_row_edge(rc, (int)((tk >> 24) & 7u), s0, xmax), ER = rec_row_edge(rc, (int)((tk >> 27) & 7u), s0, xmax);
          const int c0 = max(EL.ix1, obx0), c1 = min(ER.ix2, obx0 + obw - 1);
          for (int c = c0; c <= c1; c++) {
            const int v = sft::row_edge_area(EL, c) - sft::row_edge_area(ER, c);
            if (v) acc_add(acc, arow + c - obx0, v);
          }
        }
      }
      if (__ballot(on && (tk >> 31))) {  // a row with overlapping quads: every source of the object (sf_tor.h), one by one
        if (on && (tk >> 31)) {
          const unsigned multi = multi_sources(okind);
          // the quads through the row (at most four: the rows' pass has seen to it), their slots and their cells at the row's top / the next row's
          unsigned aq = 0u;  // per quad a byte: k | ls << 4 ... (k < 16, slots < 8: k | ls << 4 in one byte, rs apart)
          unsigned ars = 0u;
          int nact = 0;
          for (int k = 0; k < nq && nact < 4; k++) {
            const uint32_t* rc = C.rec(q0 + k);
            if (!((int)rc[16] <= s0 && (int)rc[17] >= s0 + sft::kGridY)) continue;
            int ls, rs;
            row_slots(rc, row, &ls, &rs);
            if (ls < 0 || rs < 0) continue;
            aq |= ((unsigned)k | ((unsigned)ls << 4)) << (8 * nact);
            ars |= (unsigned)rs << (8 * nact);
            nact++;
          }
          const double sd0 = (double)s0, sd1 = (double)(s0 + sft::kGridY);
          auto cells = [&](int k, int sl, int* t, int* b) {
            const uint32_t* rc = C.rec(q0 + k);
            const unsigned bwd = rc[40 + sl];
            if (s0 >= (int)(bwd & 4095u) && s0 < (int)((bwd >> 12) & 4095u)) { *t = *b = (bwd >> 24) ? xmax : 0; return; }
            *t = cell_fast(rc + slot_at(sl), sd0);
            *b = cell_fast(rc + slot_at(sl), sd1);
          };
          for (int si = 0; si < nact + 4; si++) {
            unsigned members = 0u;
            int sign = 1;
            if (si >= nact) {
              members = (multi >> (4 * (si - nact))) & 15u;
              if (!members) continue;
              sign = (__popc(members) & 1) ? 1 : -1;
            }
            int lq = -1, rq = -1, lt = 0, lb = 0, rt = 0, rb = 0, ls = 0, rs = 0, cn = 0;
            for (int a = 0; a < nact; a++) {
              const int k = (int)((aq >> (8 * a)) & 15u), als = (int)((aq >> (8 * a + 4)) & 15u), ars_a = (int)((ars >> (8 * a)) & 15u);
              if (si < nact) { if (a != si) continue; }
              else {
                const int sl = slot_of(okind, k, nq, m0);
                if (sl < 0 || !((members >> sl) & 1u)) continue;
              }
              cn++;
              int a_lt, a_lb, a_rt, a_rb;
              cells(k, als, &a_lt, &a_lb);
              cells(k, ars_a, &a_rt, &a_rb);
              if (lq < 0 || a_lt > lt || (a_lt == lt && a_lb > lb)) { lq = k; lt = a_lt; lb = a_lb; ls = als; }
              if (rq < 0 || a_rt < rt || (a_rt == rt && a_rb < rb)) { rq = k; rt = a_rt; rb = a_rb; rs = ars_a; }
            }
            if (cn != (si < nact ? 1 : __popc(members)) || lt > rt) continue;
            const sft::RowEdge EL = rec_row_edge(C.rec(q0 + lq), ls, s0, xmax), ER = rec_row_edge(C.rec(q0 + rq), rs, s0, xmax);
            const int c0 = max(EL.ix1, obx0), c1 = min(ER.ix2, obx0 + obw - 1);
            for (int c = c0; c <= c1; c++) {
              const int v = sft::row_edge_area(EL, c) - sft::row_edge_area(ER, c);
              if (v) acc_add(acc, arow + c - obx0, sign * v);
            }
          }
        }
      }
    }
    C.sync();
  }
  if (SFTD_STOP == 2) return;
  // ---- the quads' runs of sub-rows (their owners' lanes), NOW that the rows taken whole are known: a quad's sub-rows in such rows
  // have nothing to add, and for a stroke they are the middle of it -- the rows between the two at its ends that hold its corners,
  // a quarter to a half of its sub-rows.  Where a quad's whole rows are one block (always, for a stroke; anything else keeps one run
  // and the per-sub-row test below) its run is cut in two, [s_lo, the block) and [behind the block, s_hi): counts, scans, the
  // headers the sub-rows' lanes read -- first all the upper parts, then all the lower ones.
#ifndef SFTD_SPLIT_RUNS
#define SFTD_SPLIT_RUNS 1
#endif
  int cnt = 0, s_lo = 0, cnt_b = 0, s_b = 0;
  if (valid) {
    s_lo = max(my_lo, 0);
    const int s_hi = min(my_hi, C.H * sft::kGridY);
    cnt = max(s_hi - s_lo, 0);
    if (SFTD_SPLIT_RUNS && cnt > 0) {
      const uint32_t* ob = C.obj(oi);
      const int oby0 = (int)((ob[0] >> 8) & 255u);
      const int r0 = ((s_lo * 34953) >> 19) - oby0, r1 = (((s_hi - 1) * 34953) >> 19) - oby0;  // the quad's rows within its object
      if (r0 >= 0 && r1 < 32) {
        const unsigned m = (ob[6] >> r0) & ((2u << (r1 - r0)) - 1u);  // its rows that are taken whole
        if (m) {
          const int fw = __builtin_ctz(m), lw = 31 - __builtin_clz(m);
          if (m == (((2u << (lw - fw)) - 1u) << fw)) {
            const int a_end = (oby0 + r0 + fw) * sft::kGridY;
            s_b = (oby0 + r0 + lw + 1) * sft::kGridY;
            cnt = max(a_end - s_lo, 0);
            cnt_b = max(s_hi - s_b, 0);
          }
        }
      }
    }
  }
  const int incl = wave_incl_sum(cnt), incl_b = wave_incl_sum(cnt_b);
  const int tot_a = __builtin_amdgcn_readlane(incl, 63), tot_sub = tot_a + __builtin_amdgcn_readlane(incl_b, 63);
  const int start = incl - cnt, start_b = tot_a + incl_b - cnt_b;
  if (valid) {  // the quad's extent in pixel columns, [lo, hi) clamped to a byte each: for the overlap test of the sub-rows' lanes
    const int lo = min(min(mine.x[0], mine.x[1]), min(mine.x[2], mine.x[3])), hi = max(max(mine.x[0], mine.x[1]), max(mine.x[2], mine.x[3]));
    C.hdr(qi)[5] = (uint32_t)(min(max(lo >> 8, 0), 255) | (min(max((hi + 255) >> 8, 0), 255) << 8));
  }
#pragma unroll
  for (int i = 0; i < kMapWordsF; i += 64) C.map()[i + lane] = 0u;  // (the task list's words: the rows are done with them)
  C.sync();
  // (which quad a run belongs to: the r-th run is the r-th (quad, part) that has sub-rows: its header sits at r)
  const unsigned long long runs = __ballot(valid && cnt > 0), runs_b = __ballot(valid && cnt_b > 0);
  if (valid && (cnt > 0 || cnt_b > 0)) {
    const uint32_t* ob = C.obj(oi);
    const int obx0 = (int)(ob[0] & 255u), oby0 = (int)((ob[0] >> 8) & 255u), obw = (int)((ob[0] >> 16) & 255u);
    const int q0 = qi - kq, okind = kind & 255, m0 = kind >> 8, nq = nq_mine, k = kq;
    // the earlier quads this one may overlap (at most two, and whether all three can meet): by kind
    int p1 = 255, p2 = 255, triple = 0;
    if (okind == kKindLines3) { p1 = k >= 1 ? q0 : 255; p2 = k == 2 ? q0 + 1 : 255; triple = k == 2; }
    else if (okind == kKindShell) { p1 = k >= 1 ? q0 + k - 1 : 255; p2 = k == 3 ? q0 : 255; }
    else if (okind == kKindRing) { p1 = k == m0 ? q0 + m0 - 1 : (k == nq - 1 ? q0 : 255); }
    // (the partners' pixel columns: word 5 of the header slot with the partner's NUMBER, written before the barrier above)
    const int px1 = p1 != 255 ? (int)C.hdr(p1)[5] : 0, px2 = p2 != 255 ? (int)C.hdr(p2)[5] : 0;
    const int4 common{0, (int)ob[3] - oby0 * obw, obx0 | (obw << 8) | (oby0 << 16) | (oi << 24), p1 | (p2 << 8) | (triple << 16) | (qi << 24)};
    const uint32_t pxw = (uint32_t)(px1 & 0xffff) | ((uint32_t)(px2 & 0xffff) << 16);
    if (cnt > 0) {
      const int rk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(runs >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)runs, 0u));
      uint32_t* h = C.hdr(rk);
      *reinterpret_cast<int4*>(h) = int4{start | (s_lo << 16), common.y, common.z, common.w};
      h[4] = pxw;
    }
    if (cnt_b > 0) {
      const int rk = (int)__popcll(runs) + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(runs_b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)runs_b, 0u));
      uint32_t* h = C.hdr(rk);
      *reinterpret_cast<int4*>(h) = int4{start_b | (s_b << 16), common.y, common.z, common.w};
      h[4] = pxw;
    }
  }
  C.sync();
  // ---- sub-rows of the other rows: a lane per (quad, sub-row of the quad).  The lane adds its quad's span; what the quad shares
  // with the EARLIER quads it may overlap is taken off again by the same lane: -(pairs) +(the triple): the union of the object's
  // quads by inclusion-exclusion.  Which quad a lane's sub-row belongs to: a bit map of the runs' starts (a v_mbcnt pair and a
  // small table), kMapWordsF words of it per round.
  {
    constexpr int kRound = 32 * kMapWordsF;
    for (int p0 = 0; p0 < tot_sub; p0 += kRound) {
      if (p0 > 0) {
#pragma unroll
        for (int i = 0; i < kMapWordsF; i += 64) C.map()[i + lane] = 0u;
        C.sync();
      }
      // (a run's start t > 0 is marked by the bit of t - 1: lane t's count of the bits below it is the run's number)
      const bool marks = valid && cnt > 0 && start > 0, marks_b = valid && cnt_b > 0 && start_b > 0;
      if (marks && start - 1 >= p0 && start - 1 < p0 + kRound)
        __hip_atomic_fetch_or(C.map() + ((start - 1 - p0) >> 5), 1u << ((start - 1 - p0) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (marks_b && start_b - 1 >= p0 && start_b - 1 < p0 + kRound)
        __hip_atomic_fetch_or(C.map() + ((start_b - 1 - p0) >> 5), 1u << ((start_b - 1 - p0) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int kb = (int)__popcll(__ballot(marks && start - 1 < p0)) + (int)__popcll(__ballot(marks_b && start_b - 1 < p0));
      C.sync();
      const int end = min(tot_sub, p0 + kRound);
      for (int base = p0; base < end; base += 64) {
        const int t = base + lane;
        const uint2 mw = *reinterpret_cast<const uint2*>(C.map() + ((base - p0) >> 5));
        const int rk = kb + (int)__builtin_amdgcn_mbcnt_hi(mw.y, __builtin_amdgcn_mbcnt_lo(mw.x, 0u));
        kb += __popc(mw.x) + __popc(mw.y);
        const bool have = t < tot_sub;
        const uint32_t* hp = C.hdr(have ? rk : 0);
        const int4 hd = *reinterpret_cast<const int4*>(hp);
        const unsigned pxr = hp[4];
        const int q = (int)((unsigned)hd.w >> 24);
        const uint32_t* rc = C.rec(q);
        const int s = (int)((unsigned)hd.x >> 16) + t - (int)(hd.x & 0xffff);
        const int obx0 = hd.z & 255, obw = (hd.z >> 8) & 255, oby0 = (hd.z >> 16) & 255, o = (int)((unsigned)hd.z >> 24);
        const int row = (s * 34953) >> 19;
        const bool live = have && (row - oby0 >= 32 || !((C.obj(o)[6] >> (row - oby0)) & 1u));
        const int arow = hd.y + row * obw;
        const double sd = (double)s;
        int L, R;
        span_fast(rc, s, sd, live, xmax, &L, &R);
        if (live) add_span_fast(acc, arow, obx0, obw, L, R, 1);
        const int p1 = hd.w & 255, p2 = (hd.w >> 8) & 255;
        // (a partner whose pixel columns the span does not reach shares nothing with it: most of a quad's sub-rows)
        const bool near1 = R > (int)((pxr & 255u) << 8) && L < (int)(((pxr >> 8) & 255u) << 8);
        const bool near2 = R > (int)(((pxr >> 16) & 255u) << 8) && L < (int)((pxr >> 24) << 8);
        const bool want = live && R > L && ((p1 != 255 && near1) || (p2 != 255 && near2));
        if (__ballot(want)) {
          int L1 = 1, R1 = 0, L2 = 1, R2 = 0;
          const bool want1 = want && p1 != 255 && near1;
          const uint32_t* r1 = C.rec(want1 ? p1 : q);
          const int2 rg1 = *reinterpret_cast<const int2*>(r1 + 16);
          const bool in1 = want1 && s >= rg1.x && s < rg1.y;
          if (__ballot(in1)) {
            span_fast(r1, s, sd, in1, xmax, &L1, &R1);
            L1 = max(L1, L); R1 = min(R1, R);
            if (in1) add_span_fast(acc, arow, obx0, obw, L1, R1, -1);
            else { L1 = 1; R1 = 0; }
          }
          const bool want2 = want && p2 != 255 && near2;
          if (__ballot(want2)) {
            const uint32_t* r2 = C.rec(want2 ? p2 : q);
            const int2 rg2 = *reinterpret_cast<const int2*>(r2 + 16);
            const bool in2 = want2 && s >= rg2.x && s < rg2.y;
            if (__ballot(in2)) {
              span_fast(r2, s, sd, in2, xmax, &L2, &R2);
              L2 = max(L2, L); R2 = min(R2, R);
              if (in2) {
                add_span_fast(acc, arow, obx0, obw, L2, R2, -1);
                if ((hd.w >> 16) & 1) add_span_fast(acc, arow, obx0, obw, max(L1, L2), min(R1, R2), 1);
              }
            }
          }
        }
      }
      C.sync();
    }
  }
  if (SFTD_STOP == 3) return;
  if (PHASE == 1) {
    *carry = RasterCarry{nobj, tot_pix};
    return;
  }
  raster_fast_pixels(C, nobj, tot_pix);
}

// ---- pixels: object after object (the reference composites its strokes in order; boxes of different objects may overlap);
// then the accumulators are zero again for the next call
__device__ __forceinline__ void raster_fast_pixels(const CtxF& C, int nobj, int tot_pix) {
  const int lane = C.lane;
  uint32_t* const acc = C.acc();
  for (int o = 0; o < nobj; o++) {
    const uint32_t* ob = C.obj(o);
    const int obx0 = (int)(ob[0] & 255u), oby0 = (int)((ob[0] >> 8) & 255u), obw = (int)((ob[0] >> 16) & 255u), obh = (int)(ob[0] >> 24);
    const int ab = (int)ob[3], ogrey = (int)(ob[4] >> 24), n = obw * obh;
    const float rw = __builtin_amdgcn_rcpf((float)obw);
    for (int i = lane; i < n; i += 64) {
      const int ry = (int)(((float)i + 0.5f) * rw), rx = i - ry * obw;
      const int cov = (int)((acc[(ab + i) >> 1] >> (16 * ((ab + i) & 1))) & 0xffffu);
      const int a = sft::area_to_alpha(cov);
      if (a) {
        uint8_t* p = C.fb + (oby0 + ry) * C.W + obx0 + rx;
        *p = (uint8_t)sft::lerp8(ogrey, a, *p);
      }
    }
    C.sync();
  }
  for (int i = lane; i < (tot_pix + 1) / 2; i += 64) acc[i] = 0u;
  C.sync();
}

}  // namespace sftd
