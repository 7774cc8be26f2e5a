// sf_tor.h -- how cairo 1.16's image backend turns a stroked polygon into 8-bit coverage, in the form the frame kernels
// evaluate it: shared by the HIP kernels (sf_render.hip) and the host's picture tables (sf_image.cpp), one source.
//
// The reference draws every frame with cairo (SRC/draw.cpp:82-270; python/spacefortress/setup.py:8 links the system's).  Its
// pixels are therefore cairo's rasterisation of the reference's paths: 24.8 fixed-point vertices, a scan converter that samples
// 15 sub-rows per pixel row with x rounded to 1/256 pixel at each sub-row's centre, a shortcut for pixel rows in which no edge
// starts, ends or crosses another (the edge's exact trapezoid, its column crossings quantised to whole sub-rows), coverage to
// alpha by (17 c + 256) >> 9, and an 8-bit lerp per pixel (cairo-tor-scan-converter.c, cairo-path-stroke-polygon.c,
// cairo-image-compositor.c of cairo 1.16.0).  oracle/cairo_model.c restates that pipeline edge list by edge list, pinned bit for
// bit to the real library; THIS file is the product's formulation of the same arithmetic for what the frame kernels draw --
// convex quads (a wireframe's stroke is a parallelogram, an explosion's arc piece a quad) united per object -- arranged for
// lanes, with cairo's 64-bit integer quotients and remainders carried as integer-valued doubles (exact below 2^53; every
// division is corrected to the exact floor):
//
//   stroke_quad      the four corners of one butt-capped line segment (compute_face: the half-width offset comes from the
//                    FIXED-POINT slope and is itself rounded to 1/256)
//   edge_ab          cell(s) = floor(A + B s): the edge's x at the centre of sub-row s, rounded half up to 1/256 (polygon_add_edge
//                    / step / edge->cell); one FMA per evaluation, a bias of a quarter of the smallest possible distance to a
//                    rounding boundary makes the float form agree with the rational one (exhaustively compared on the host)
//   quad_interval    [max over left edges, min over right edges] of the cells: the span sub_row() would emit for the quad alone
//   full_row_edge    cell_list_render_edge: the area right of an edge over a whole pixel row, cairo's way
//   row_is_full      can_do_full_row + the bucket test: no vertex strictly inside the row, no two edges change order
//   source           a set of quads of one object; its interval is the INTERSECTION of theirs, its sign (-1)^(n-1): the union
//                    of an object's strokes (one cairo_stroke = one polygon, non-zero winding) by inclusion-exclusion
//   area_to_alpha, lerp8   GRID_AREA_TO_ALPHA and _fill_xrgb32_lerp_opaque_spans' arithmetic
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SFT_HD __host__ __device__ __forceinline__
#else
#define SFT_HD inline
#endif

namespace sft {

constexpr int kGridY = 15;
constexpr int kFull = 2 * 256 * kGridY;  // a fully covered pixel: 7680

// ---- fixed point ----------------------------------------------------------------------------------------------------------
SFT_HD int fx_from_double(double d) {  // _cairo_fixed_from_double: round to nearest (ties to even) at 2^-8
  const double m = d + 26388279066624.0;  // 1.5 * 2^44: the sum's low word is the fixed-point value
#ifdef __HIP_DEVICE_COMPILE__
  return __double2loint(m);
#else
  union { double d; int32_t i[2]; } u;
  u.d = m;
  return u.i[0];
#endif
}
SFT_HD int to_grid_y(int y) { return (kGridY * y + 128) >> 8; }  // INPUT_TO_GRID_Y: the nearest sub-row boundary

// exact floor(a / b) for integer-valued doubles, b > 0, |a| < 2^52
SFT_HD double floor_div(double a, double b) {
  double q = floor(a / b);
  const double r = fma(-q, b, a);
  if (r < 0.0) q -= 1.0;
  else if (r >= b) q += 1.0;
  return q;
}
// 1 / d where a few ulp do not matter (every use below states its margin): on the device v_rcp_f64 and two Newton steps instead
// of the IEEE division's expansion
SFT_HD double rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
#else
  return 1.0 / d;
#endif
}
SFT_HD double trunc_div(double a, double b) { return a >= 0.0 ? floor_div(a, b) : -floor_div(-a, b); }  // C's integer division

// ---- coverage -> pixel ---------------------------------------------------------------------------------------------------
SFT_HD int area_to_alpha(int c) {
  const int a = (c + (c << 4) + 256) >> 9;
  return a > 255 ? 255 : (a < 0 ? 0 : a);
}
SFT_HD int mul8(int a, int b) {  // cairo-image-compositor.c mul8x2_8 on one channel: + 0x7f, not pixman's + 0x80
  const int t = a * b + 0x7f;
  return ((t + (t >> 8)) >> 8) & 0xff;
}
SFT_HD int lerp8(int src, int a, int dst) {
  if (a == 255) return src;
  const int t = mul8(src, a) + mul8(dst, 255 - a);
  return t > 255 ? 255 : t;
}
// boxes (cairo-rectangular-scan-converter.c): area in 1/65536 of a pixel -> c = area >> 8, alpha = c - (c >> 8)
SFT_HD int box_area_to_alpha(long long area) {
  const int c = (int)(area >> 8);
  return c - (c >> 8);
}

// ---- a convex quad in 24.8 fixed point ------------------------------------------------------------------------------------
struct Quad {
  int x[4], y[4];
};

struct Affine {  // the matrix a path point goes through: x' = xx x + xy y + x0, y' = yx x + yy y + y0 (cairo_matrix_t)
  double xx, yx, xy, yy, x0, y0;
};
// drawGameStateScaled's scale + translate (SRC/draw.cpp:259-260) then drawWireFrame's translate(pos) rotate(angle)
// (:85-86), multiplied the way cairo's gstate multiplies them (cairo_matrix_multiply, new transform on the LEFT):
//   scale:      xx = sx, yy = sy
//   translate:  x0 = tx * xx + ty * xy + x0  (here xy = yx = 0)
//   rotate:     xx' = c xx + s xy, yx' = c yx + s yy, xy' = -s xx + c xy, yy' = -s yx + c yy
SFT_HD Affine view_matrix(double sx, double sy, double vpx, double vpy) {
  Affine m;
  m.xx = sx; m.yx = 0.0; m.xy = 0.0; m.yy = sy;
  m.x0 = (-vpx) * m.xx + (-vpy) * m.xy + 0.0;
  m.y0 = (-vpx) * m.yx + (-vpy) * m.yy + 0.0;
  return m;
}
SFT_HD Affine object_matrix(const Affine& v, double px, double py, double cs, double sn) {
  Affine t = v;  // translate(px, py)
  t.x0 = px * v.xx + py * v.xy + v.x0;
  t.y0 = px * v.yx + py * v.yy + v.y0;
  Affine r;      // rotate: cairo_matrix_init_rotate = (c, s, -s, c)
  r.xx = cs * t.xx + sn * t.xy;
  r.yx = cs * t.yx + sn * t.yy;
  r.xy = -sn * t.xx + cs * t.xy;
  r.yy = -sn * t.yx + cs * t.yy;
  r.x0 = 0.0 * t.xx + 0.0 * t.xy + t.x0;
  r.y0 = 0.0 * t.yx + 0.0 * t.yy + t.y0;
  return r;
}
SFT_HD void to_device(const Affine& m, double x, double y, int* fx, int* fy) {  // cairo_matrix_transform_point, then to fixed
  const double dx = m.xx * x + m.xy * y, dy = m.yx * x + m.yy * y;
  *fx = fx_from_double(dx + m.x0);
  *fy = fx_from_double(dy + m.y0);
}

// compute_face's offset for a segment whose fixed-point slope is (dx, dy): half the line width `hw` (user units) across the
// slope, through the inverse matrix and back (cairo-path-stroke-polygon.c).  For a view scale (sx, sy) under any rotation that
// is S rot90(normalize(S^-1 d)) hw -- the rotation commutes with the quarter turn -- up to the last bits of a double, and only
// the rounding of the result to 1/256 is used.
SFT_HD void face_offset(int dx, int dy, double sx, double sy, double hw, int* ox, int* oy) {
  double ux, uy;
  if (dx == 0) { ux = 0.0; uy = dy > 0 ? 1.0 : -1.0; }
  else if (dy == 0) { uy = 0.0; ux = dx > 0 ? 1.0 : -1.0; }
  else {
    const double fdx = (double)dx / sx, fdy = (double)dy / sy;
    const double mag = sqrt(fdx * fdx + fdy * fdy);
    ux = fdx / mag; uy = fdy / mag;
  }
  *ox = fx_from_double(-uy * hw * sx);
  *oy = fx_from_double(ux * hw * sy);
}
// one butt-capped segment p1 -> p2 (fixed): corners p1 + off, p2 + off, p2 - off, p1 - off (ccw side first)
SFT_HD Quad stroke_quad(int x1, int y1, int x2, int y2, double sx, double sy, double hw) {
  int ox, oy;
  face_offset(x2 - x1, y2 - y1, sx, sy, hw, &ox, &oy);
  Quad q;
  q.x[0] = x1 + ox; q.y[0] = y1 + oy;
  q.x[1] = x2 + ox; q.y[1] = y2 + oy;
  q.x[2] = x2 - ox; q.y[2] = y2 - oy;
  q.x[3] = x1 - ox; q.y[3] = y1 - oy;
  return q;
}

// ---- an edge: cell(s) = floor(A + B s) --------------------------------------------------------------------------------------
struct EdgeAB {
  double A, B;
};
// the line through (x1, y1) - (x2, y2), y1 != y2 (fixed); sub-row s's centre is y = (2 s + 1) * 256 / 30
SFT_HD EdgeAB edge_ab(int x1, int y1, int x2, int y2) {
  if (y2 < y1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
  const double dx = (double)(x2 - x1), dy30 = 30.0 * (double)(y2 - y1);
  const double K = dx / dy30;
  EdgeAB e;
  e.B = 512.0 * K;
  // + 1/2: round half up; + a quarter of the least distance 1 / (2 * 30 dy) a non-tie can have from the next integer
  e.A = (double)x1 + (256.0 - 30.0 * (double)y1) * K + 0.5 + 0.25 / (2.0 * dy30);
  return e;
}
SFT_HD int edge_cell(const EdgeAB& e, int s) { return (int)floor(fma(e.B, (double)s, e.A)); }
// the same from the rational (host checks): x1 + floor((2 N + D) / (2 D)), N = ((2s+1) 256 - 30 y1) dx, D = 30 dy
inline long long edge_cell_exact(int x1, int y1, int x2, int y2, int s) {
  if (y2 < y1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
  const long long N = ((2LL * s + 1) * 256 - 30LL * y1) * (x2 - x1), D = 30LL * (y2 - y1);
  long long num = 2 * N + D, den = 2 * D, q = num / den;
  if ((num % den) < 0) q--;
  return x1 + q;
}

// ---- a quad prepared for the scan: left and right edges, sub-row range ---------------------------------------------------------
struct QuadScan {
  EdgeAB e[4];      // edge k: vertex k -> k + 1
  unsigned left;    // bit k: edge k bounds the quad on the left (else, if not horizontal, on the right)
  unsigned horiz;   // bit k: edge k is horizontal (dropped by the polygon: no cells)
  int s0, s1;       // sub-rows [s0, s1)
  int gy[4];        // the vertices' sub-row boundaries (to_grid_y)
  // Where an edge leaves the surface sideways cairo's polygon runs along the border instead (cairo-polygon.c:
  // _add_clipped_edge): edge k is the vertical x = out_x[k] in the sub-rows [out_s0[k], out_s1[k]) -- empty for nearly every
  // edge --, and the end of that range inside the edge is one more vertex of the polygon (an event for a row's mode).
  int out_s0[4], out_s1[4], out_x[4];
};
// _cairo_edge_compute_intersection_y_for_x / x_for_y: `_cairo_fixed_mul_div_floor` is a plain C division (towards zero)
SFT_HD int line_y_for_x(int ax, int ay, int bx, int by, int x) {
  if (x == ax) return ay;
  if (x == bx) return by;
  return ay + (int)(((long long)(x - ax) * (by - ay)) / (bx - ax));
}
SFT_HD int line_x_for_y(int ax, int ay, int bx, int by, int y) {
  if (y == ay) return ax;
  if (y == by) return bx;
  return ax + (int)(((long long)(y - ay) * (bx - ax)) / (by - ay));
}
SFT_HD QuadScan quad_scan(const Quad& q, int xmax = 1 << 30) {  // xmax = the surface's width in 1/256 pixel
  QuadScan s;
  s.left = 0u; s.horiz = 0u;
  const long long cx4 = (long long)q.x[0] + q.x[1] + q.x[2] + q.x[3], cy4 = (long long)q.y[0] + q.y[1] + q.y[2] + q.y[3];
  int lo = 1 << 30, hi = -(1 << 30);
  for (int k = 0; k < 4; k++) {
    const int j = (k + 1) & 3;
    s.gy[k] = to_grid_y(q.y[k]);
    lo = s.gy[k] < lo ? s.gy[k] : lo;
    hi = s.gy[k] > hi ? s.gy[k] : hi;
    s.out_s0[k] = s.out_s1[k] = 0; s.out_x[k] = 0;
    if (q.y[k] == q.y[j]) {
      s.horiz |= 1u << k;
      s.e[k].A = 0.0; s.e[k].B = 0.0;
      continue;
    }
    s.e[k] = edge_ab(q.x[k], q.y[k], q.x[j], q.y[j]);
    // orient the edge downwards; the centroid (times 4) lies to its right iff cross < 0: then the edge is a LEFT boundary
    int x1 = q.x[k], y1 = q.y[k], x2 = q.x[j], y2 = q.y[j];
    if (y2 < y1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    const long long cross = (long long)(x2 - x1) * (cy4 - 4LL * y1) - (long long)(y2 - y1) * (cx4 - 4LL * x1);
    if (cross < 0) s.left |= 1u << k;
    // the border the edge crosses, if any (an object is far smaller than the surface: at most one)
    const int pl = x1 < x2 ? x1 : x2, pr = x1 < x2 ? x2 : x1;
    const bool down_right = x1 <= x2;  // (x1, y1) is the upper end
    if (pr <= 0 || pl >= xmax) {  // wholly beside the surface: the border itself, top to bottom
      s.out_x[k] = pr <= 0 ? 0 : xmax;
      s.out_s0[k] = to_grid_y(y1); s.out_s1[k] = to_grid_y(y2);
    } else if (pl < 0 && pr > 0) {
      int y = line_y_for_x(x1, y1, x2, y2, 0);
      if (line_x_for_y(x1, y1, x2, y2, y) < 0) y += down_right ? 1 : -1;
      y = y < y1 ? y1 : (y > y2 ? y2 : y);
      s.out_x[k] = 0;
      if (down_right) { s.out_s0[k] = to_grid_y(y1); s.out_s1[k] = to_grid_y(y); }
      else { s.out_s0[k] = to_grid_y(y); s.out_s1[k] = to_grid_y(y2); }
    } else if (pl < xmax && pr > xmax) {
      int y = line_y_for_x(x1, y1, x2, y2, xmax);
      if (line_x_for_y(x1, y1, x2, y2, y) > xmax) y += down_right ? -1 : 1;
      y = y < y1 ? y1 : (y > y2 ? y2 : y);
      s.out_x[k] = xmax;
      if (down_right) { s.out_s0[k] = to_grid_y(y); s.out_s1[k] = to_grid_y(y2); }
      else { s.out_s0[k] = to_grid_y(y1); s.out_s1[k] = to_grid_y(y); }
    }
  }
  s.s0 = lo; s.s1 = hi;
  return s;
}
constexpr int kCellMin = -(1 << 28), kCellMax = 1 << 28;
// the quad's span in sub-row s (s0 <= s < s1): [L, R) in 1/256 pixel; empty if R <= L
SFT_HD void quad_interval(const QuadScan& q, int s, int* L, int* R) {
  int l = kCellMin, r = kCellMax;
  for (int k = 0; k < 4; k++) {
    if ((q.horiz >> k) & 1u) continue;
    int c = edge_cell(q.e[k], s);
    if (s >= q.out_s0[k] && s < q.out_s1[k]) c = q.out_x[k];
    if ((q.left >> k) & 1u) l = c > l ? c : l;
    else r = c < r ? c : r;
  }
  *L = l; *R = r;
}

// ---- a whole pixel row at once (cell_list_render_edge) ------------------------------------------------------------------------
// The area to the RIGHT of the edge inside pixel column c of the row that starts at sub-row s0 (= 15 * row), in units of
// 1 / 7680 pixel, for every column: 0 left of the edge, 7680 right of it, cairo's trapezoids where it passes.
struct RowEdge {
  int ix1, fx1, ix2, fx2;  // the edge at the row's top and bottom (after the half-sub-row step back), left end first
  double X1, U, DX, rDX;   // the multi-column case: top x as a numerator over U = 30 dy, the run (x2 - x1) * U and 1 / run
  int single;              // ix1 == ix2
};
SFT_HD RowEdge row_edge(int x1, int y1, int x2, int y2, int s0) {
  if (y2 < y1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
  RowEdge r;
  const double dx = (double)(x2 - x1);
  double q1, r1, q2, r2, U = 30.0 * (double)(y2 - y1);
  if (x2 == x1) {  // vertical: no stepping at all
    q1 = q2 = (double)x1; r1 = r2 = 0.0;
  } else {
    // x at the centres of sub-rows s0 and s0 + 15 as quotient and remainder over U (the remainder in units of 1/256 of cairo's)
    const double T0 = ((double)(2 * s0 + 1) * 256.0 - 30.0 * (double)y1) * dx;
    const double T1 = T0 + 15.0 * 512.0 * dx;
    q1 = floor_div(T0, U); r1 = T0 - q1 * U; q1 += (double)x1;
    q2 = floor_div(T1, U); r2 = T1 - q2 * U; q2 += (double)x1;
    // back by half a sub-row: dxdy = 512 dx / U as C quotient and remainder, each halved on its own (the quotient truncates)
    const double Q = trunc_div(512.0 * dx, U), R = 512.0 * dx - Q * U;
    const double hq = Q >= 0.0 ? floor(Q * 0.5) : -floor(-Q * 0.5), hr = R * 0.5;
    q1 -= hq; r1 -= hr;
    if (r1 < 0.0) { q1 -= 1.0; r1 += U; } else if (r1 >= U) { q1 += 1.0; r1 -= U; }
    q2 -= hq; r2 -= hr;
    if (r2 < 0.0) { q2 -= 1.0; r2 += U; } else if (r2 >= U) { q2 += 1.0; r2 -= U; }
  }
  int a = (int)q1, b = (int)q2;
  r.ix1 = a >> 8; r.fx1 = a & 255; r.ix2 = b >> 8; r.fx2 = b & 255;
  r.single = r.ix1 == r.ix2;
  if (r.ix2 < r.ix1) {
    int t = r.ix1; r.ix1 = r.ix2; r.ix2 = t;
    t = r.fx1; r.fx1 = r.fx2; r.fx2 = t;
    double d = q1; q1 = q2; q2 = d;
    d = r1; r1 = r2; r2 = d;
  }
  r.U = U;
  r.X1 = q1 * U + r1;
  r.DX = (q2 - q1) * U + (r2 - r1);
  r.rDX = r.single ? 0.0 : rcp(r.DX);
  return r;
}
// The same WITHOUT a division, from the edge's A + B s (edge_ab) -- what the lanes use.  x(s) is a rational with denominator
// U = 30 dy: where it is not a whole number it lies at least 1 / U from one, and the float form is good to 1e-10: quotient
// floor(x + 1 / (2 U)), remainder round((x - quotient) U), both exact; likewise the step 512 dx / U = B itself.
// (dx, dy: the edge's run in fixed point, dy > 0 after orienting it downwards; x1: its upper end's x; vertical edges: dx == 0.)
SFT_HD RowEdge row_edge_ab(const EdgeAB& e, int dx, int dy, int x1, int s0) {
  RowEdge r;
  const double U = 30.0 * (double)dy, hU = 0.5 * rcp(U);  // (the half steps below: margins of 1 / (2 U) against 1e-10)
  double q1, r1, q2, r2;
  if (dx == 0) {
    q1 = q2 = (double)x1; r1 = r2 = 0.0;
  } else {
    const double off = 0.5 + 0.25 * hU;  // what edge_ab folded into A
    const double xa = fma(e.B, (double)s0, e.A) - off, xb = fma(e.B, (double)(s0 + kGridY), e.A) - off;
    q1 = floor(xa + hU); r1 = rint((xa - q1) * U);
    q2 = floor(xb + hU); r2 = rint((xb - q2) * U);
    const double Q = e.B >= 0.0 ? floor(e.B + hU) : -floor(-e.B + hU), R = rint((e.B - Q) * U);
    const double hq = Q >= 0.0 ? floor(Q * 0.5) : -floor(-Q * 0.5), hr = R * 0.5;
    q1 -= hq; r1 -= hr;
    if (r1 < 0.0) { q1 -= 1.0; r1 += U; } else if (r1 >= U) { q1 += 1.0; r1 -= U; }
    q2 -= hq; r2 -= hr;
    if (r2 < 0.0) { q2 -= 1.0; r2 += U; } else if (r2 >= U) { q2 += 1.0; r2 -= U; }
  }
  int a = (int)q1, b = (int)q2;
  r.ix1 = a >> 8; r.fx1 = a & 255; r.ix2 = b >> 8; r.fx2 = b & 255;
  r.single = r.ix1 == r.ix2;
  if (r.ix2 < r.ix1) {
    int t = r.ix1; r.ix1 = r.ix2; r.ix2 = t;
    t = r.fx1; r.fx1 = r.fx2; r.fx2 = t;
    double d = q1; q1 = q2; q2 = d;
    d = r1; r1 = r2; r2 = d;
  }
  r.U = U;
  r.X1 = q1 * U + r1;
  r.DX = (q2 - q1) * U + (r2 - r1);
  r.rDX = r.single ? 0.0 : rcp(r.DX);
  return r;
}
// whole sub-rows the edge needs to reach column boundary 256 c (multi-column case), cairo's quotient stepping = exact floors
// (the quotient is 0 .. 15 and its numerator and denominator are whole: off a whole number by 1 / DX at least)
SFT_HD int row_edge_y(const RowEdge& e, int c) {
  const double t = ((double)c * 256.0 * e.U - e.X1) * 15.0;
  return (int)floor((t + 0.5) * e.rDX);
}
SFT_HD int row_edge_area(const RowEdge& e, int c) {
  if (c < e.ix1) return 0;
  if (c > e.ix2) return kFull;
  if (e.single) return kGridY * (512 - e.fx1 - e.fx2);
  if (c == e.ix1) return row_edge_y(e, c + 1) * (256 - e.fx1);
  if (c == e.ix2) {
    const int y = row_edge_y(e, c);
    return 512 * y + (kGridY - y) * (512 - e.fx2);
  }
  return 256 * (row_edge_y(e, c + 1) + row_edge_y(e, c));
}


// ---- drawExplosion (SRC/draw.cpp:116-145): arcs as cairo flattens and strokes them ----------------------------------------
// cairo_arc(xc, yc, r, a1, a2) for a2 - a1 <= pi with one segment (cairo-arc.c: every arc here needs one: tolerance 0.1 over a
// radius of at most 63 * .2 device pixels allows pi / 2) is ONE Bezier curve; its four control points in user space are
// xc / yc plus these eight numbers, which depend on (r, a1, a2) alone -- made once on the host with libm, like cairo does:
struct ArcK {
  double rca, rsa, hrsa, hrca, rcb, rsb, hrsb, hrcb;  // r cos A, r sin A, h r sin A, h r cos A, and the same at B; h = 4/3 tan((B - A) / 4)
};
SFT_HD ArcK arc_k(double r, double A, double B) {  // (called on the host: the table is libm's, like cairo's own values)
  ArcK k;
  k.rsa = r * sin(A); k.rca = r * cos(A);
  k.rsb = r * sin(B); k.rcb = r * cos(B);
  const double h = 4.0 / 3.0 * tan((B - A) / 4.0);
  k.hrsa = h * k.rsa; k.hrca = h * k.rca; k.hrsb = h * k.rsb; k.hrcb = h * k.rcb;
  return k;
}
struct Knots {
  int ax, ay, bx, by, cx, cy, dx, dy;  // fixed
};
SFT_HD Knots arc_knots(const Affine& m, double xc, double yc, const ArcK& k) {
  Knots s;
  to_device(m, xc + k.rca, yc + k.rsa, &s.ax, &s.ay);                        // cairo_arc's line_to (= move_to) to the start
  to_device(m, xc + k.rca - k.hrsa, yc + k.rsa + k.hrca, &s.bx, &s.by);    // _cairo_arc_segment
  to_device(m, xc + k.rcb + k.hrsb, yc + k.rsb - k.hrcb, &s.cx, &s.cy);
  to_device(m, xc + k.rcb, yc + k.rsb, &s.dx, &s.dy);
  return s;
}
// _de_casteljau: the first (half = 0) or second half of the curve, in place (arithmetic shifts, like cairo's fixed halving)
SFT_HD Knots spline_half(const Knots& s, int half) {
  const int abx = (s.ax + s.bx) >> 1, aby = (s.ay + s.by) >> 1, bcx = (s.bx + s.cx) >> 1, bcy = (s.by + s.cy) >> 1;
  const int cdx = (s.cx + s.dx) >> 1, cdy = (s.cy + s.dy) >> 1;
  const int abbcx = (abx + bcx) >> 1, abbcy = (aby + bcy) >> 1, bccdx = (bcx + cdx) >> 1, bccdy = (bcy + cdy) >> 1;
  const int fx = (abbcx + bccdx) >> 1, fy = (abbcy + bccdy) >> 1;
  Knots r;
  if (half == 0) { r.ax = s.ax; r.ay = s.ay; r.bx = abx; r.by = aby; r.cx = abbcx; r.cy = abbcy; r.dx = fx; r.dy = fy; }
  else { r.ax = fx; r.ay = fy; r.bx = bccdx; r.by = bccdy; r.cx = cdx; r.cy = cdy; r.dx = s.dx; r.dy = s.dy; }
  return r;
}
// _cairo_spline_error_squared: how far the control points are from the chord (the flattening stops below tolerance^2 = 0.01)
SFT_HD double spline_error_sq(const Knots& k) {
  double bdx = (double)(k.bx - k.ax) / 256.0, bdy = (double)(k.by - k.ay) / 256.0;
  double cdx = (double)(k.cx - k.ax) / 256.0, cdy = (double)(k.cy - k.ay) / 256.0;
  if (k.ax != k.dx || k.ay != k.dy) {
    const double dx = (double)(k.dx - k.ax) / 256.0, dy = (double)(k.dy - k.ay) / 256.0, v = dx * dx + dy * dy;
    double u = bdx * dx + bdy * dy;
    if (u <= 0) {
    } else if (u >= v) { bdx -= dx; bdy -= dy; }
    else { bdx -= u / v * dx; bdy -= u / v * dy; }
    u = cdx * dx + cdy * dy;
    if (u <= 0) {
    } else if (u >= v) { cdx -= dx; cdy -= dy; }
    else { cdx -= u / v * dx; cdy -= u / v * dy; }
  }
  const double be = bdx * bdx + bdy * bdy, ce = cdx * cdx + cdy * cdy;
  return be > ce ? be : ce;
}
// the quad between two faces of a stroked curve (spline_to appends face.cw / face.ccw to the two contours): the face at
// (px, py) with tangent (tx, ty), and the next one -- corners in stroke_quad's order (ccw side first)
SFT_HD Quad faces_quad(int p0x, int p0y, int t0x, int t0y, int p1x, int p1y, int t1x, int t1y, double sx, double sy, double hw) {
  int o0x, o0y, o1x, o1y;
  face_offset(t0x, t0y, sx, sy, hw, &o0x, &o0y);
  face_offset(t1x, t1y, sx, sy, hw, &o1x, &o1y);
  Quad q;
  q.x[0] = p0x + o0x; q.y[0] = p0y + o0y;
  q.x[1] = p1x + o1x; q.y[1] = p1y + o1y;
  q.x[2] = p1x - o1x; q.y[2] = p1y - o1y;
  q.x[3] = p0x - o0x; q.y[3] = p0y - o0y;
  return q;
}
// a 10-degree arc of the explosion: one piece (its control points lie 0.07 pixel from the chord at most), start face along
// a -> b, end face along c -> d
SFT_HD Quad arc_quad_fixed(const Knots& s, double sx, double sy, double hw) {
  return faces_quad(s.ax, s.ay, s.bx - s.ax, s.by - s.ay, s.dx, s.dy, s.dx - s.cx, s.dy - s.cy, sx, sy, hw);
}
// piece i (0..7) of a half circle flattened three levels deep, as _cairo_spline_decompose emits it: the face at the piece's
// start (tangent a -> b of the piece; the curve's own initial slope for piece 0) and at its end (the next piece's start; the
// curve's final slope c -> d for piece 7)
SFT_HD Quad ring_piece_quad(const Knots& half, int i, double sx, double sy, double hw) {
  Knots p = spline_half(spline_half(spline_half(half, (i >> 2) & 1), (i >> 1) & 1), i & 1);
  int t1x, t1y;
  if (i == 7) { t1x = half.dx - half.cx; t1y = half.dy - half.cy; }
  else {
    const int j = i + 1;
    const Knots n = spline_half(spline_half(spline_half(half, (j >> 2) & 1), (j >> 1) & 1), j & 1);
    t1x = n.bx - n.ax; t1y = n.by - n.ay;
  }
  // (the first face is made from the curve's own initial slope a -> b, not from its first leaf's)
  const int t0x = i == 0 ? half.bx - half.ax : p.bx - p.ax, t0y = i == 0 ? half.by - half.ay : p.by - p.ay;
  return faces_quad(p.ax, p.ay, t0x, t0y, p.dx, p.dy, t1x, t1y, sx, sy, hw);
}

// ---- the general case: _cairo_spline_decompose with tolerance 0.1 (cairo-spline.c) ---------------------------------------------
// The faces of a stroked curve (curve_to / spline_to of cairo-path-stroke-polygon.c): face 0 at the curve's start along its
// initial slope, then one at the start of every leaf of the adaptive subdivision but the first (tangent a -> b of the leaf;
// skipped when the point repeats), the last at the curve's end along its final slope.  px / py / tx / ty hold up to `cap`
// faces; returns their number (n faces = n - 1 quads).  In the default geometry every 10-degree arc is one leaf and a half
// circle eight (arc_quad_fixed, ring_piece_quad: the frame kernel's fixed forms); larger scales subdivide further.
SFT_HD int flatten_faces(const Knots& k, int* px, int* py, int* tx, int* ty, int cap) {
  int n = 0;
  px[n] = k.ax; py[n] = k.ay; tx[n] = k.bx - k.ax; ty[n] = k.by - k.ay; n++;
  if (tx[0] == 0 && ty[0] == 0) { tx[0] = k.cx - k.ax; ty[0] = k.cy - k.ay; }
  Knots st[8];
  int sp = 0;
  st[sp++] = k;
  int lastx = k.ax, lasty = k.ay;
  while (sp > 0) {
    const Knots c = st[--sp];
    if (spline_error_sq(c) < 0.01 || sp >= 6) {
      if (!(c.ax == lastx && c.ay == lasty) && n < cap - 1) {
        px[n] = c.ax; py[n] = c.ay; tx[n] = c.bx - c.ax; ty[n] = c.by - c.ay; n++;
        lastx = c.ax; lasty = c.ay;
      }
    } else {
      st[sp++] = spline_half(c, 1);  // (the first half is decomposed first)
      st[sp++] = spline_half(c, 0);
    }
  }
  px[n] = k.dx; py[n] = k.dy; tx[n] = k.dx - k.cx; ty[n] = k.dy - k.cy; n++;
  return n;
}

}  // namespace sft
