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
  __device__ __forceinline__ uint32_t* rec(int q) const { return lds + q * kRecWords; }
  __device__ __forceinline__ uint32_t* obj(int o) const { return lds + maxq * kRecWords + o * kObjWords; }
  __device__ __forceinline__ uint32_t* acc() const { return lds + maxq * kRecWords + kMaxObjs * kObjWords; }
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
template <int MAXACT = 4>
__device__ __forceinline__ void raster(const Ctx& C, const sft::Quad& mine, bool valid, int obj0, int kind, int grey) {
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
  // ---- the objects' boxes, in one lane each (lane o < nobj), then prefix sums over the objects
  int bx0 = 0, by0 = 0, bw = 0, bh = 0, S0 = 0, nsub = 0;
  if (lane < nobj) {
    const int* o = reinterpret_cast<const int*>(C.obj(lane));
    const int x0 = max(o[0] >> 8, 0), x1 = min((o[2] + 255) >> 8, C.W);
    const int s0 = max(o[1], 0), s1 = min(o[3], C.H * sft::kGridY);
    if (x1 > x0 && s1 > s0) {
      bx0 = x0; bw = x1 - x0;
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
    const uint32_t nq = o[5];
    o[0] = (uint32_t)bx0 | ((uint32_t)by0 << 8) | ((uint32_t)bw << 16) | ((uint32_t)bh << 24);
    o[1] = (uint32_t)S0 | (o[7] << 16);    // (+ a circle's first half in the upper bits)
    o[2] = (uint32_t)nsub;
    o[3] = (uint32_t)(abase - bw * bh);         // where the object's accumulator starts (pixels)
    o[4] |= nq << 8;
    o[5] = (uint32_t)(rbase - bh);             // where its rows start in the enumeration of rows
    o[6] = 0u;                                  // its rows' modes (bit r: row by0 + r is taken whole)
    o[7] = (uint32_t)(sbase - nsub);           // where its sub-rows start
  }
  C.sync();
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
}

}  // namespace sftd
