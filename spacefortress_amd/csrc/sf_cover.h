// sf_cover.h -- exact area coverage of a convex quad (a stroke: a line with butt caps, an arc's chord, a filled rectangle)
// over a pixel, as the frame kernels composite it (sf_render.hip, sf_render_generic.hip): the anti-aliasing model of the
// image observation (sf_raster.h says what is and is not pinned about it).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

namespace sfcov {

struct Quad {
  float x[4], y[4];
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// mean over t in [0,1] of clamp(ya + t*(yb-ya), 0, 1).  Branch-free on purpose: an `if` on per-pixel data is an
// exec-mask save / branch / restore (six scalar instructions and two hand-offs) around a dozen vector ones, four
// times per pixel; both forms are evaluated and one is selected -- the same arithmetic on the path taken, so the
// same pixels (the discarded form may hold an inf or a NaN: a select does not look at it).
__device__ __forceinline__ float ramp_mean(float ya, float yb) {
  const float lo = fminf(ya, yb), d = fabsf(yb - ya);  // the mean does not depend on the direction
  const float flat = clamp01(lo + 0.5f * d);
  const float inv = __builtin_amdgcn_rcpf(d);  // 1 ulp: moves a coverage by 1e-7, far below one grey level
  const float ta = clamp01(-lo * inv), tb = clamp01((1.0f - lo) * inv);
  const float ramp = (1.0f - tb) + (tb - ta) * (lo + 0.5f * d * (ta + tb));
  return d < 1e-6f ? flat : ramp;
}

// one directed edge's share of the integral of clamp(y, 0, 1) dx over the pixel at the origin; `slope` = dy / dx of
// the edge: a property of the stroke, not of the pixel (quad_slopes) -- computed per pixel it was four reciprocals
// and a dozen instructions of the 165 a pixel costs
__device__ __forceinline__ float edge_term(float x0, float y0, float x1, float slope) {
  const float xa = clamp01(x0), xb = clamp01(x1);
  const float w = xb - xa;
  const float t = w * ramp_mean(y0 + (xa - x0) * slope, y0 + (xb - x0) * slope);
  return w == 0.f ? 0.f : t;  // (a vertical edge: slope inf or NaN, w = 0)
}
__device__ __forceinline__ float edge_slope(float x0, float y0, float x1, float y1) {
  return (y1 - y0) * __builtin_amdgcn_rcpf(x1 - x0);
}
struct Slopes {
  float s[4];
};
__device__ __forceinline__ Slopes quad_slopes(const Quad& q) {
  Slopes sl;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int f = (e + 1) & 3;
    sl.s[e] = edge_slope(q.x[e], q.y[e], q.x[f], q.y[f]);
  }
  return sl;
}

// ... and for TWO edges at once, on pairs of floats: gfx950 has packed float32 multiply / add / subtract (v_pk_mul_f32,
// v_pk_add_f32: two IEEE operations per instruction, each rounded exactly as its scalar form -- nothing is contracted or
// reordered), and a pixel's four edge terms are the same two dozen operations on different operands.  The kernel's bound
// is the NUMBER of instructions issued: the products, sums and differences of a pair of edges issue once instead of
// twice (min / max / clamp / reciprocal have no packed form and stay per edge) -- a third of the 150 vector instructions
// a pixel cost.  Same operations on the same operands in the same order per edge: the same coverage, bit for bit.
#ifndef SF_PK_COVER
#define SF_PK_COVER 1
#endif
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2_t clamp01(f2_t v) { return f2_t{clamp01(v.x), clamp01(v.y)}; }
__device__ __forceinline__ f2_t ramp_mean(f2_t ya, f2_t yb) {
  const f2_t lo = f2_t{fminf(ya.x, yb.x), fminf(ya.y, yb.y)};
  const f2_t dd = yb - ya;
  const f2_t d = f2_t{fabsf(dd.x), fabsf(dd.y)};
  const f2_t flat = clamp01(lo + 0.5f * d);
  const f2_t inv = f2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2_t ta = clamp01(-lo * inv), tb = clamp01((1.0f - lo) * inv);
  const f2_t ramp = (1.0f - tb) + (tb - ta) * (lo + 0.5f * d * (ta + tb));
  return f2_t{d.x < 1e-6f ? flat.x : ramp.x, d.y < 1e-6f ? flat.y : ramp.y};
}
__device__ __forceinline__ f2_t edge_term(f2_t x0, f2_t y0, f2_t x1, f2_t slope) {
  const f2_t xa = clamp01(x0), xb = clamp01(x1);
  const f2_t w = xb - xa;
  const f2_t t = w * ramp_mean(y0 + (xa - x0) * slope, y0 + (xb - x0) * slope);
  return f2_t{w.x == 0.f ? 0.f : t.x, w.y == 0.f ? 0.f : t.y};
}

// area of quad /\ pixel [px,px+1]x[py,py+1]:  | sum over edges of the integral of clamp(y,0,1) dx |
__device__ __forceinline__ float quad_cover(const Quad& q, const Slopes& sl, float px, float py) {
  float s = 0.f;
  if (SF_PK_COVER) {
    const f2_t pp = {px, px}, qq = {py, py};
    const f2_t x01 = f2_t{q.x[0], q.x[1]} - pp, x23 = f2_t{q.x[2], q.x[3]} - pp;
    const f2_t y01 = f2_t{q.y[0], q.y[1]} - qq, y23 = f2_t{q.y[2], q.y[3]} - qq;
    const f2_t t01 = edge_term(x01, y01, f2_t{x01.y, x23.x}, f2_t{sl.s[0], sl.s[1]});
    const f2_t t23 = edge_term(x23, y23, f2_t{x23.y, x01.x}, f2_t{sl.s[2], sl.s[3]});
    s += t01.x;
    s += t01.y;
    s += t23.x;
    s += t23.y;
    return fabsf(s);
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int f = (e + 1) & 3;
    s += edge_term(q.x[e] - px, q.y[e] - py, q.x[f] - px, sl.s[e]);
  }
  return fabsf(s);
}
__device__ __forceinline__ float quad_cover(const Quad& q, float px, float py) {  // (callers whose quad is fixed over their
  return quad_cover(q, quad_slopes(q), px, py);                                   //  pixel loop: the compiler hoists the slopes)
}

}  // namespace sfcov
