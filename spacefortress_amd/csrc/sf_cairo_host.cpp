// sf_cairo_host.cpp -- host side of the image observation's rasterisation (no HIP calls; tested without a GPU).
//
//  1. What only the host draws: the two hexagons (a closed path with miter joins: SRC/draw.cpp:102-114) and the vulnerability
//     bar's rectangles (:207-225), for any geometry -- a general polygon through cairo's scan conversion (sweep_polygon) and the
//     box converter (boxes_cover).  Restated from cairo 1.16's published algorithm (cairo-path-stroke-polygon.c: compute_face,
//     outer_join, inner_join; cairo-tor-scan-converter.c; cairo-rectangular-scan-converter.c); the arithmetic primitives are
//     sf_tor.h's.
//  2. The frame kernels' formulation -- an object as convex quads united by inclusion-exclusion, a pixel row either sampled in
//     15 sub-rows or taken whole (sf_tor.h) -- run on the host, lane loops as plain loops (object_coverage): the pictures the
//     kernels copy instead of drawing (the fortress at its 36 headings, its explosion) come from here, and tests compare it
//     with oracle/cairo_model.c on arbitrary poses before anything runs on a GPU (sf_image_object_alpha).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "sf_cairo_host.h"
#include "sf_internal.h"
#include "sf_raster.h"

namespace sfh {

using sft::Quad;

// ================= 1. general polygons (host only) ===========================================================================
namespace {

struct Pt { int x, y; };
struct PEdge { Pt p1, p2; int top, bottom, dir; };  // cairo_edge_t: the line p1 -> p2 (p1 above), clipped to [top, bottom)

struct Face { Pt ccw, point, cw; int sdx, sdy; double ux, uy; };

int trunc_muldiv(int a, int b, int c) { return (int)(((long long)a * b) / c); }  // _cairo_fixed_mul_div_floor: C division
int edge_x_for_y(const Pt& a, const Pt& b, int y) {
  if (y == a.y) return a.x;
  if (y == b.y) return b.x;
  return a.x + ((b.y - a.y) ? trunc_muldiv(y - a.y, b.x - a.x, b.y - a.y) : 0);
}
int edge_y_for_x(const Pt& a, const Pt& b, int x) {
  if (x == a.x) return a.y;
  if (x == b.x) return b.y;
  return a.y + ((b.x - a.x) ? trunc_muldiv(x - a.x, b.y - a.y, b.x - a.x) : 0);
}

struct Polygon {
  std::vector<PEdge> e;
  Pt l1, l2;  // limits: the surface (cairo-polygon.c: _add_clipped_edge)
  void push(const Pt& a, const Pt& b, int top, int bottom, int dir) {
    if (top < bottom) e.push_back(PEdge{a, b, top, bottom, dir});
  }
  void add(Pt a, Pt b, int dir) {  // _cairo_polygon_add_edge + the clipping to the limits
    if (a.y == b.y) return;
    if (a.y > b.y) { std::swap(a, b); dir = -dir; }
    if (b.y <= l1.y || a.y >= l2.y) return;
    const Pt bl{l1.x, l2.y}, tr{l2.x, l1.y};
    int top_y = std::max(a.y, l1.y), bot_y = std::min(b.y, l2.y);
    const int pl = std::min(a.x, b.x), pr = std::max(a.x, b.x);
    if (l1.x <= pl && pr <= l2.x) { push(a, b, top_y, bot_y, dir); return; }
    if (pr <= l1.x) { push(l1, bl, top_y, bot_y, dir); return; }
    if (l2.x <= pl) { push(tr, l2, top_y, bot_y, dir); return; }
    int left_y, right_y;
    if ((a.x <= b.x) == (a.y <= b.y)) {
      if (pl >= l1.x) left_y = top_y;
      else { left_y = edge_y_for_x(a, b, l1.x); if (edge_x_for_y(a, b, left_y) < l1.x) left_y++; }
      left_y = std::min(left_y, bot_y);
      if (top_y < left_y) { push(l1, bl, top_y, left_y, dir); top_y = left_y; }
      if (pr <= l2.x) right_y = bot_y;
      else { right_y = edge_y_for_x(a, b, l2.x); if (edge_x_for_y(a, b, right_y) > l2.x) right_y--; }
      right_y = std::max(right_y, top_y);
      if (bot_y > right_y) { push(tr, l2, right_y, bot_y, dir); bot_y = right_y; }
    } else {
      if (pr <= l2.x) right_y = top_y;
      else { right_y = edge_y_for_x(a, b, l2.x); if (edge_x_for_y(a, b, right_y) > l2.x) right_y++; }
      right_y = std::min(right_y, bot_y);
      if (top_y < right_y) { push(tr, l2, top_y, right_y, dir); top_y = right_y; }
      if (pl >= l1.x) left_y = bot_y;
      else { left_y = edge_y_for_x(a, b, l1.x); if (edge_x_for_y(a, b, left_y) < l1.x) left_y--; }
      left_y = std::max(left_y, top_y);
      if (bot_y > left_y) { push(l1, bl, left_y, bot_y, dir); bot_y = left_y; }
    }
    if (top_y != bot_y) push(a, b, top_y, bot_y, dir);
  }
  void contour(const std::vector<Pt>& c, int dir) {  // _cairo_polygon_add_contour: an open chain
    for (size_t i = 1; i < c.size(); i++) add(c[i - 1], c[i], dir);
  }
};

Face make_face(const Pt& p, int sdx, int sdy, double sx, double sy, double hw) {
  Face f;
  int ox, oy;
  sft::face_offset(sdx, sdy, sx, sy, hw, &ox, &oy);
  f.ccw = Pt{p.x + ox, p.y + oy};
  f.point = p;
  f.cw = Pt{p.x - ox, p.y - oy};
  f.sdx = sdx; f.sdy = sdy;
  double dx = sdx / 256.0, dy = sdy / 256.0;  // dev_slope: normalize_slope
  if (sdx == 0) { f.ux = 0; f.uy = sdy > 0 ? 1 : -1; }
  else if (sdy == 0) { f.uy = 0; f.ux = sdx > 0 ? 1 : -1; }
  else { const double m = hypot(dx, dy); f.ux = dx / m; f.uy = dy / m; }
  return f;
}

// a closed polygon stroked with miter joins (limit 10): the two contours of cairo-path-stroke-polygon.c
void stroke_closed(const std::vector<Pt>& pts, double sx, double sy, double hw, Polygon* poly) {
  const int n = (int)pts.size();
  std::vector<Pt> cw, ccw;
  Face first{}, cur{};
  bool have = false;
  auto sgn = [](double a, double b, double c, double d) { const double v = a * d - c * b; return v > 0 ? 1 : (v < 0 ? -1 : 0); };
  auto join = [&](const Face& in, const Face& out) {
    const long long cr = (long long)in.sdx * out.sdy - (long long)out.sdx * in.sdy;
    if (cr == 0) return;
    const bool clockwise = cr > 0;  // the turn is towards the ccw side: cw is the outside
    std::vector<Pt>& outer = clockwise ? cw : ccw;
    std::vector<Pt>& inner = clockwise ? ccw : cw;
    const Pt& inpt = clockwise ? in.cw : in.ccw;
    const Pt& outpt = clockwise ? out.cw : out.ccw;
    bool mitered = false;
    if (!(in.cw.x == out.cw.x && in.cw.y == out.cw.y && in.ccw.x == out.ccw.x && in.ccw.y == out.ccw.y)) {
      const double dot = in.ux * out.ux + in.uy * out.uy;
      if (2 <= 100.0 * (1 + dot)) {
        const double x1 = inpt.x / 256.0, y1 = inpt.y / 256.0, dx1 = in.ux, dy1 = in.uy;
        const double x2 = outpt.x / 256.0, y2 = outpt.y / 256.0, dx2 = out.ux, dy2 = out.uy;
        const double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
        const double mx = fabs(dy1) >= fabs(dy2) ? (my - y1) * dx1 / dy1 + x1 : (my - y2) * dx2 / dy2 + x2;
        const double ix = in.point.x / 256.0, iy = in.point.y / 256.0;
        if (sgn(x1 - ix, y1 - iy, mx - ix, my - iy) != sgn(x2 - ix, y2 - iy, mx - ix, my - iy)) {
          const Pt p{sft::fx_from_double(mx), sft::fx_from_double(my)};
          outer.back() = p;
          outer.front() = p;
          mitered = true;
        }
      }
      if (!mitered) outer.push_back(outpt);
    }
    inner.push_back(in.point);
    inner.push_back(clockwise ? out.ccw : out.cw);
  };
  auto line_to = [&](const Pt& from, const Pt& to) {
    if (from.x == to.x && from.y == to.y) return;
    const int sdx = to.x - from.x, sdy = to.y - from.y;
    Face start = make_face(from, sdx, sdy, sx, sy, hw);
    if (have) join(cur, start);
    else { first = start; have = true; cw.push_back(start.cw); ccw.push_back(start.ccw); }
    cur = start;
    cur.point = to;
    cur.ccw.x += sdx; cur.ccw.y += sdy;
    cur.cw.x += sdx; cur.cw.y += sdy;
    cw.push_back(cur.cw);
    ccw.push_back(cur.ccw);
  };
  for (int i = 0; i < n; i++) line_to(pts[i], pts[(i + 1) % n]);
  join(cur, first);
  poly->contour(cw, 1);
  poly->contour(ccw, -1);
}

// ---- cairo-tor-scan-converter.c on an edge list: non-zero winding, coverage (0..7680) per pixel of the window ---------------
struct SEdge {
  int x1, y1, x2, y2;  // the line, top point first
  int ytop, ybot, dir;
  int cell;            // at the current sub-row
  long long cell_at(int s) const { return x1 == x2 ? x1 : sft::edge_cell_exact(x1, y1, x2, y2, s); }
};

void sweep_polygon(const std::vector<PEdge>& pe, int W, int H, std::vector<int>* cover) {
  cover->assign((size_t)W * H, 0);
  std::vector<SEdge> ed;
  for (const PEdge& p : pe) {
    SEdge e;
    e.ytop = std::max(sft::to_grid_y(p.top), 0);
    e.ybot = std::min(sft::to_grid_y(p.bottom), H * sft::kGridY);
    if (e.ybot <= e.ytop) continue;
    if (p.p2.y > p.p1.y) { e.x1 = p.p1.x; e.y1 = p.p1.y; e.x2 = p.p2.x; e.y2 = p.p2.y; e.dir = p.dir; }
    else { e.x1 = p.p2.x; e.y1 = p.p2.y; e.x2 = p.p1.x; e.y2 = p.p1.y; e.dir = -p.dir; }
    e.cell = 0;
    ed.push_back(e);
  }
  if (ed.empty()) return;
  std::vector<int> act;  // the active list, in cairo's list order
  auto merge_new = [&](int s) {
    std::vector<int> nw;
    for (int i = 0; i < (int)ed.size(); i++)
      if (ed[i].ytop == s) { ed[i].cell = (int)ed[i].cell_at(s); nw.push_back(i); }
    if (nw.empty()) return;
    std::stable_sort(nw.begin(), nw.end(), [&](int a, int b) { return ed[a].cell < ed[b].cell; });
    size_t pos = 0;
    for (int i : nw) {  // merge_sorted_edges: list elements go first on equal cells
      while (pos < act.size() && ed[act[pos]].cell <= ed[i].cell) pos++;
      act.insert(act.begin() + pos, i);
      pos++;
    }
  };
  auto add_span = [&](int row, int x1, int x2, int unit) {  // [x1, x2) in 1/256 pixel, `unit` = 2 per sub-row
    if (x2 <= x1) return;
    for (int px = std::max(x1 >> 8, 0); px <= std::min((x2 - 1) >> 8, W - 1); px++) {
      const int lo = std::max(x1, px << 8), hi = std::min(x2, (px + 1) << 8);
      if (hi > lo) (*cover)[(size_t)row * W + px] += unit * (hi - lo);
    }
  };
  for (int row = 0; row < H; row++) {
    const int s0 = row * sft::kGridY;
    bool starts_inside = false;
    for (const SEdge& e : ed) starts_inside |= e.ytop > s0 && e.ytop < s0 + sft::kGridY;
    bool full = false;
    if (!starts_inside) {
      merge_new(s0);
      if (act.empty()) continue;
      full = true;
      int prev = INT32_MIN;
      for (int i : act) {
        if (ed[i].ybot - s0 < sft::kGridY) { full = false; break; }
        const int c = (int)ed[i].cell_at(s0 + sft::kGridY);
        if (c < prev) { full = false; break; }
        prev = c;
      }
    }
    if (full) {  // full_row: spans by winding; a span that ends where the next begins runs on
      size_t i = 0;
      while (i < act.size()) {
        const int li = act[i];
        int winding = ed[li].dir;
        size_t j = i + 1;
        for (;; j++) {
          winding += ed[act[j]].dir;
          const int nextcell = j + 1 < act.size() ? ed[act[j + 1]].cell : INT32_MAX;
          if (winding == 0 && nextcell != ed[act[j]].cell) break;
        }
        const SEdge& L = ed[li];
        const SEdge& R = ed[act[j]];
        const sft::RowEdge rl = sft::row_edge(L.x1, L.y1, L.x2, L.y2, s0), rr = sft::row_edge(R.x1, R.y1, R.x2, R.y2, s0);
        for (int px = 0; px < W; px++)
          (*cover)[(size_t)row * W + px] += sft::row_edge_area(rl, px) - sft::row_edge_area(rr, px);
        i = j + 1;
      }
      std::vector<int> keep;
      for (int k : act)
        if (ed[k].ybot > s0 + sft::kGridY) { ed[k].cell = (int)ed[k].cell_at(s0 + sft::kGridY); keep.push_back(k); }
      act.swap(keep);
    } else {
      for (int sub = 0; sub < sft::kGridY; sub++) {
        const int s = s0 + sub;
        if (!(sub == 0 && !starts_inside)) merge_new(s);
        int xstart = INT32_MIN, winding = 0;
        for (size_t i = 0; i < act.size(); i++) {  // sub_row: spans in list order
          const SEdge& e = ed[act[i]];
          winding += e.dir;
          if (winding == 0) {
            const int nextcell = i + 1 < act.size() ? ed[act[i + 1]].cell : INT32_MAX;
            if (nextcell != e.cell) { add_span(row, xstart, e.cell, 2); xstart = INT32_MIN; }
          } else if (xstart == INT32_MIN) {
            xstart = e.cell;
          }
        }
        std::vector<int> nx;  // step: drop what ends here, move the rest, insertion-sort by the new cells
        for (int k : act) {
          if (ed[k].ybot <= s + 1) continue;
          ed[k].cell = (int)ed[k].cell_at(s + 1);
          size_t pos = nx.size();
          while (pos > 0 && ed[nx[pos - 1]].cell > ed[k].cell) pos--;
          nx.insert(nx.begin() + pos, k);
        }
        act.swap(nx);
      }
    }
  }
}

void composite(const std::vector<int>& cover, int W, int H, int grey, uint8_t* fb) {
  for (int i = 0; i < W * H; i++) {
    const int a = sft::area_to_alpha(cover[i]);
    if (a) fb[i] = (uint8_t)sft::lerp8(grey, a, fb[i]);
  }
}

}  // namespace

void stroke_hexagon(const double* pts12, const Geometry& g, int grey, uint8_t* fb) {
  const sft::Affine v = sft::view_matrix(g.sx, g.sy, g.vp_x, g.vp_y);
  std::vector<Pt> p(6);
  for (int i = 0; i < 6; i++) sft::to_device(v, pts12[2 * i], pts12[2 * i + 1], &p[i].x, &p[i].y);
  Polygon poly;
  poly.l1 = Pt{0, 0};
  poly.l2 = Pt{g.w * 256, g.h * 256};
  stroke_closed(p, g.sx, g.sy, g.lw / 2.0, &poly);
  std::vector<int> cover;
  sweep_polygon(poly.e, g.w, g.h, &cover);
  composite(cover, g.w, g.h, grey, fb);
}

// the union of axis-aligned boxes (fixed point), exact area per pixel, the box converter's alpha
void boxes_cover(const Box4* bx, int nb, int W, int H, int grey, uint8_t* fb) {
  int minx = INT32_MAX, miny = INT32_MAX, maxx = INT32_MIN, maxy = INT32_MIN;
  for (int i = 0; i < nb; i++) {
    minx = std::min(minx, bx[i].x1); miny = std::min(miny, bx[i].y1);
    maxx = std::max(maxx, bx[i].x2); maxy = std::max(maxy, bx[i].y2);
  }
  for (int py = std::max(miny >> 8, 0); py < std::min((maxy + 255) >> 8, H); py++)
    for (int px = std::max(minx >> 8, 0); px < std::min((maxx + 255) >> 8, W); px++) {
      std::vector<int> xs{px * 256, px * 256 + 256}, ys{py * 256, py * 256 + 256};
      for (int i = 0; i < nb; i++) {
        for (int v : {bx[i].x1, bx[i].x2}) if (v > xs[0] && v < xs[1]) xs.push_back(v);
        for (int v : {bx[i].y1, bx[i].y2}) if (v > ys[0] && v < ys[1]) ys.push_back(v);
      }
      std::sort(xs.begin(), xs.end());
      std::sort(ys.begin(), ys.end());
      long long area = 0;
      for (size_t a = 0; a + 1 < xs.size(); a++)
        for (size_t b = 0; b + 1 < ys.size(); b++) {
          if (xs[a] == xs[a + 1] || ys[b] == ys[b + 1]) continue;
          bool in = false;
          for (int i = 0; i < nb && !in; i++)
            in = bx[i].x1 <= xs[a] && xs[a + 1] <= bx[i].x2 && bx[i].y1 <= ys[b] && ys[b + 1] <= bx[i].y2;
          if (in) area += (long long)(xs[a + 1] - xs[a]) * (ys[b + 1] - ys[b]);
        }
      const int al = sft::box_area_to_alpha(area);
      if (al) fb[py * W + px] = (uint8_t)sft::lerp8(grey, al, fb[py * W + px]);
    }
}

// cairo_rectangle(x, y, w, h) + cairo_fill under the view matrix: move_to the corner, the sides as rounded DISTANCES
Box4 user_rect(const Geometry& g, double x, double y, double w, double h) {
  const sft::Affine v = sft::view_matrix(g.sx, g.sy, g.vp_x, g.vp_y);
  int x0, y0;
  sft::to_device(v, x, y, &x0, &y0);
  const int dw = sft::fx_from_double(v.xx * w + v.xy * 0.0), dh = sft::fx_from_double(v.yx * 0.0 + v.yy * h);
  Box4 b{x0, y0, x0 + dw, y0 + dh};
  if (b.x2 < b.x1) std::swap(b.x1, b.x2);
  if (b.y2 < b.y1) std::swap(b.y1, b.y2);
  return b;
}

// ================= 2. the kernels' formulation, on the host =====================================================================
// accumulate the object's coverage (0..7680 per pixel) into acc[H][W]
void object_coverage(const Object& ob, int W, int H, int* acc) {
  const int nq = ob.nq;
  std::vector<sft::QuadScan> qs(nq);
  int S0 = INT32_MAX, S1 = INT32_MIN;
  for (int k = 0; k < nq; k++) {
    qs[k] = sft::quad_scan(ob.q[k], W * 256);
    S0 = std::min(S0, qs[k].s0);
    S1 = std::max(S1, qs[k].s1);
  }
  S0 = std::max(S0, 0);
  S1 = std::min(S1, H * sft::kGridY);
  if (S1 <= S0) return;
  for (int row = S0 / sft::kGridY; row <= (S1 - 1) / sft::kGridY; row++) {
    const int s0 = row * sft::kGridY;
    // ---- the row's mode (can_do_full_row and the buckets): a vertex strictly inside the row, or two edges that swap places
    bool full = true;
    for (int k = 0; k < nq && full; k++) {
      for (int v = 0; v < 4; v++) full &= !(qs[k].gy[v] > s0 && qs[k].gy[v] < s0 + sft::kGridY);
      for (int e = 0; e < 4; e++) {
        if (ob.chain && ((e == 1 && k < nq - 1) || (e == 3 && k > 0))) continue;  // (no polygon edge: nothing to clip)
        for (int ev : {qs[k].out_s0[e], qs[k].out_s1[e]})
          if (qs[k].out_s1[e] > qs[k].out_s0[e]) full &= !(ev > s0 && ev < s0 + sft::kGridY);
      }
    }
    struct Act { int quad, le, re, lt, lb, rt, rb; };  // the quad's two edges through the row: cells at the top / the next row's top
    std::vector<Act> act;
    if (full) {
      // One entry per edge through the row, in the order cairo's active list has them at the row's top: by cell; equal cells keep
      // the order of one sub-row earlier (the list is re-sorted stably after every step); an edge that starts with this row
      // comes behind the ones already there, and edges that start together come in the polygon's edge order (per stroke: the
      // ccw side, the far cap, the near cap, the cw side -- add_caps).  can_do_full_row wants the cells at the next row's top
      // non-decreasing in that order.
      struct Ord { int top, is_new, tie, rank, bottom; };
      std::vector<Ord> ord;
      for (int k = 0; k < nq; k++) {
        if (!(qs[k].s0 <= s0 && qs[k].s1 >= s0 + sft::kGridY)) continue;
        Act a{k, -1, -1, 0, 0, 0, 0};
        for (int e = 0; e < 4; e++) {
          if ((qs[k].horiz >> e) & 1u) continue;
          const int g0 = std::min(qs[k].gy[e], qs[k].gy[(e + 1) & 3]), g1 = std::max(qs[k].gy[e], qs[k].gy[(e + 1) & 3]);
          if (!(g0 <= s0 && g1 >= s0 + sft::kGridY)) continue;
          if ((qs[k].left >> e) & 1u) a.le = e; else a.re = e;
        }
        if (a.le < 0 || a.re < 0) continue;
        auto cells = [&](int e, int* t, int* b) {
          static const int kRank[4] = {0, 1, 3, 2};
          const bool out = s0 >= qs[k].out_s0[e] && s0 < qs[k].out_s1[e];
          const int g0 = std::max(std::min(qs[k].gy[e], qs[k].gy[(e + 1) & 3]), 0);
          // the piece of the edge that is in the list: along the border (from out_s0) or the edge proper (from its top / out_s1)
          const int start = out ? std::max(qs[k].out_s0[e], 0) : (qs[k].out_s1[e] > qs[k].out_s0[e] && qs[k].out_s1[e] <= s0 && qs[k].out_s1[e] > g0 ? qs[k].out_s1[e] : g0);
          Ord o;
          if (out) { *t = *b = qs[k].out_x[e]; o.tie = qs[k].out_x[e]; }
          else {
            *t = sft::edge_cell(qs[k].e[e], s0);
            *b = sft::edge_cell(qs[k].e[e], s0 + sft::kGridY);
            o.tie = sft::edge_cell(qs[k].e[e], s0 - 1);
          }
          o.top = *t; o.bottom = *b; o.is_new = start == s0; o.rank = 8 * k + kRank[e] + (out ? 4 : 0);
          if (o.is_new) o.tie = o.rank;
          // (the faces between the pieces of a flattened curve bound the pieces but are no edges of cairo's polygon)
          const bool shared_face = ob.chain && ((e == 1 && k < nq - 1) || (e == 3 && k > 0));
          if (!shared_face) ord.push_back(o);
        };
        cells(a.le, &a.lt, &a.lb);
        cells(a.re, &a.rt, &a.rb);
        act.push_back(a);
      }
      std::stable_sort(ord.begin(), ord.end(), [](const Ord& x, const Ord& y) {
        if (x.top != y.top) return x.top < y.top;
        if (x.is_new != y.is_new) return x.is_new < y.is_new;
        if (x.tie != y.tie) return x.tie < y.tie;
        return x.rank < y.rank;
      });
      for (size_t i = 1; i < ord.size(); i++) full &= ord[i].bottom >= ord[i - 1].bottom;
    }
    // ---- every source of the object (single quads +, pairs -, triples +: the union)
    for (int si = 0; si < ob.nsrc; si++) {
      const unsigned members = ob.src_members[si];
      const int sign = ob.src_sign[si];
      if (full) {
        // the source's left edge: of its members' left edges the last in the list (greatest top cell; on a tie the greater
        // bottom cell); its right edge: the first of their right edges.  They overlap if left.top <= right.top (a span that
        // ends where the next begins runs on: the tie counts).
        int lq = -1, rq = -1, lt = 0, lb = 0, rt = 0, rb = 0, le = 0, re = 0, cnt = 0, want = 0;
        for (int k = 0; k < nq; k++) want += (members >> k) & 1u;
        for (const Act& a : act) {
          if (!((members >> a.quad) & 1u)) continue;
          cnt++;
          if (lq < 0 || a.lt > lt || (a.lt == lt && a.lb > lb)) { lq = a.quad; lt = a.lt; lb = a.lb; le = a.le; }
          if (rq < 0 || a.rt < rt || (a.rt == rt && a.rb < rb)) { rq = a.quad; rt = a.rt; rb = a.rb; re = a.re; }
        }
        if (cnt != want || lt > rt) continue;
        if (want > 1 && lt == rt && lq != rq) {
          // two spans that merely touch at the row's top merge only if the later one's left edge comes right behind
          // the earlier one's right edge; as a pair's "intersection" that is [left, right] with left.top == right.top: kept
        }
        const Quad& ql = ob.q[lq];
        const Quad& qr = ob.q[rq];
        auto edge_of = [&](const Quad& q, const sft::QuadScan& sc, int e) {
          if (s0 >= sc.out_s0[e] && s0 < sc.out_s1[e]) return sft::row_edge(sc.out_x[e], 0, sc.out_x[e], 256, s0);  // along the border
          // (the division-free form the lanes use; row_edge is the plain restatement, and the general polygons' sweep uses that)
          int x1 = q.x[e], y1 = q.y[e], x2 = q.x[(e + 1) & 3], y2 = q.y[(e + 1) & 3];
          if (y2 < y1) { std::swap(x1, x2); std::swap(y1, y2); }
          return sft::row_edge_ab(sc.e[e], x2 - x1, y2 - y1, x1, s0);
        };
        const sft::RowEdge L = edge_of(ql, qs[lq], le), R = edge_of(qr, qs[rq], re);
        for (int px = 0; px < W; px++) acc[row * W + px] += sign * (sft::row_edge_area(L, px) - sft::row_edge_area(R, px));
      } else {
        for (int s = std::max(s0, S0); s < std::min(s0 + sft::kGridY, S1); s++) {
          int L = sft::kCellMin, R = sft::kCellMax;
          bool on = true;
          for (int k = 0; k < nq && on; k++) {
            if (!((members >> k) & 1u)) continue;
            if (s < qs[k].s0 || s >= qs[k].s1) { on = false; break; }
            int l, r;
            sft::quad_interval(qs[k], s, &l, &r);
            L = std::max(L, l);
            R = std::min(R, r);
          }
          if (!on || R <= L) continue;
          for (int px = std::max(L >> 8, 0); px <= std::min((R - 1) >> 8, W - 1); px++) {
            const int lo = std::max(L, px << 8), hi = std::min(R, (px + 1) << 8);
            if (hi > lo) acc[row * W + px] += sign * 2 * (hi - lo);
          }
        }
      }
    }
  }
}

// ---- the reference's objects as quads -------------------------------------------------------------------------------------------
static const double kShip[3][4] = {{-18, 0, 18, 0}, {-18, 18, 0, 0}, {0, 0, -18, -18}};                        // SRC/wireframe.cpp:40-52
static const double kFort[4][4] = {{0, 0, 36, 0}, {0, -18, 18, -18}, {18, -18, 18, 18}, {18, 18, 0, 18}};      // :54-67
static const double kMissile[3][4] = {{0, 0, -25, 0}, {0, 0, -5, 5}, {0, 0, -5, -5}};                          // :11-22
static const double kShell[4][4] = {{-8, 0, 0, -6}, {0, -6, 16, 0}, {16, 0, 0, 6}, {0, 6, -8, 0}};            // :24-38

void all_subsets(Object* ob) {  // every non-empty subset of the quads: singles +, pairs -, ... (nq <= 4)
  ob->nsrc = 0;
  for (unsigned m = 1; m < (1u << ob->nq); m++) {
    ob->src_members[ob->nsrc] = m;
    ob->src_sign[ob->nsrc] = (__builtin_popcount(m) & 1) ? 1 : -1;
    ob->nsrc++;
  }
}

Object wireframe_object(int kind, double px, double py, int angle_deg, const Geometry& g, const double* cos_sin_of_deg) {
  const double(*lines)[4] = kind == 0 ? kShip : kind == 1 ? kFort : kind == 2 ? kMissile : kShell;
  const int n = (kind == 0 || kind == 2) ? 3 : 4;
  const sft::Affine v = sft::view_matrix(g.sx, g.sy, g.vp_x, g.vp_y);
  const sft::Affine m = sft::object_matrix(v, px, py, cos_sin_of_deg[0], cos_sin_of_deg[1]);
  (void)angle_deg;
  Object ob;
  ob.chain = false;
  ob.nq = n;
  for (int k = 0; k < n; k++) {
    int x1, y1, x2, y2;
    sft::to_device(m, lines[k][0], lines[k][1], &x1, &y1);
    sft::to_device(m, lines[k][2], lines[k][3], &x2, &y2);
    ob.q[k] = sft::stroke_quad(x1, y1, x2, y2, g.sx, g.sy, g.lw / 2.0);
  }
  all_subsets(&ob);
  return ob;
}


// drawExplosion (SRC/draw.cpp:116-145) at (x, y): 84 arcs -- radius 15 + 8 i, i = 0..6, twelve per ring, from 30 k + 3 (i + 1)
// degrees over 10 -- each its own cairo_stroke (one quad), grey .75 below radius 60 and .5 above; then the radius-7 circle,
// one stroke: two half circles of eight pieces each.  Composited in that order onto fb.
void draw_explosion(double x, double y, const Geometry& g, uint8_t* fb) {
  const sft::Affine v = sft::view_matrix(g.sx, g.sy, g.vp_x, g.vp_y);
  const double hw = (double)(float)g.lw / 2.0;  // `float ls`
  std::vector<int> acc((size_t)g.w * g.h);
  auto paint = [&](const Object& ob, int grey) {
    std::fill(acc.begin(), acc.end(), 0);
    object_coverage(ob, g.w, g.h, acc.data());
    for (int i = 0; i < g.w * g.h; i++) {
      const int a = sft::area_to_alpha(acc[i]);
      if (a) fb[i] = (uint8_t)sft::lerp8(grey, a, fb[i]);
    }
  };
  const double kPi = 3.14159265358979323846;
  auto d2r = [&](double a) { return a * kPi / 180; };
  int ofs = 0;
  for (int radius = 15; radius < 70; radius += 8) {
    ofs += 3;
    for (int angle = 0; angle < 360; angle += 30) {
      const sft::ArcK k = sft::arc_k((double)radius, d2r(angle + ofs), d2r(angle + ofs + 10));
      int px[6], py[6], tx[6], ty[6];
      const int n = sft::flatten_faces(sft::arc_knots(v, x, y, k), px, py, tx, ty, 5);  // (one piece at scale .2: arc_quad_fixed)
      Object ob;
      ob.chain = true;
      ob.nq = n - 1;
      for (int p = 0; p + 1 < n; p++) ob.q[p] = sft::faces_quad(px[p], py[p], tx[p], ty[p], px[p + 1], py[p + 1], tx[p + 1], ty[p + 1], g.sx, g.sy, hw);
      ob.nsrc = 0;
      for (int i = 0; i < ob.nq; i++) { ob.src_members[ob.nsrc] = 1u << i; ob.src_sign[ob.nsrc++] = 1; }
      paint(ob, radius < 60 ? 191 : 128);
    }
  }
  Object ring;
  ring.chain = true;
  ring.nq = 0;
  int m0 = 0;
  for (int h = 0; h < 2; h++) {
    // cairo_arc(0, 2 pi) is two arcs of pi (_cairo_arc_in_direction halves anything longer), one Bezier segment each
    const sft::ArcK k = sft::arc_k(7.0, h ? 0.0 + (2 * kPi - 0.0) / 2.0 : 0.0, h ? 2 * kPi : 0.0 + (2 * kPi - 0.0) / 2.0);
    int px[18], py[18], tx[18], ty[18];
    const int n = sft::flatten_faces(sft::arc_knots(v, x, y, k), px, py, tx, ty, 17);  // (eight pieces at scale .2: ring_piece_quad)
    for (int p = 0; p + 1 < n; p++) ring.q[ring.nq++] = sft::faces_quad(px[p], py[p], tx[p], ty[p], px[p + 1], py[p + 1], tx[p + 1], ty[p + 1], g.sx, g.sy, hw);
    if (h == 0) m0 = ring.nq;
  }
  // the pieces abut (shared faces) except where two faces were made from different tangents at the same point: the seam
  // between the halves and the closing one -- there the two quads may overlap by a sliver: counted once
  ring.nsrc = 0;
  for (int i = 0; i < ring.nq; i++) { ring.src_members[ring.nsrc] = 1u << i; ring.src_sign[ring.nsrc++] = 1; }
  ring.src_members[ring.nsrc] = (1u << (m0 - 1)) | (1u << m0); ring.src_sign[ring.nsrc++] = -1;
  ring.src_members[ring.nsrc] = (1u << (ring.nq - 1)) | (1u << 0); ring.src_sign[ring.nsrc++] = -1;
  paint(ring, 191);
}

}  // namespace sfh

// ---- C ABI: what tests (and table builders outside this file) call ------------------------------------------------------------
extern "C" int sf_trig_deg(int deg, double* cos_sin);  // sf_host.cpp: cos / sin of deg2rad(deg) as the reference computes them

// the 8-bit coverage (alpha) of ONE wireframe on a w x h surface, through the frame kernels' formulation on the host
extern "C" int sf_image_object_alpha(int kind, double x, double y, int angle_deg, int w, int h, double vp_x, double vp_y,
                                     double vp_w, double vp_h, double lw, uint8_t* alpha) {
  if (kind < 0 || kind > 3 || !alpha || w <= 0 || h <= 0) {
    sf_set_error("sf_image_object_alpha: bad argument");
    return SF_ERR_ARG;
  }
  const sfh::Geometry g{w, h, (double)w / vp_w, (double)h / vp_h, vp_x, vp_y, lw};
  double cs[2];
  sf_trig_deg(((angle_deg % 360) + 360) % 360, cs);
  const sfh::Object ob = sfh::wireframe_object(kind, x, y, angle_deg, g, cs);
  std::vector<int> acc((size_t)w * h, 0);
  sfh::object_coverage(ob, w, h, acc.data());
  for (int i = 0; i < w * h; i++) alpha[i] = (uint8_t)sft::area_to_alpha(acc[i]);
  return SF_OK;
}

// drawExplosion at (x, y) composited onto `fb` (w x h, the caller's background), through the kernels' formulation on the host
extern "C" int sf_image_explosion_host(double x, double y, int w, int h, double vp_x, double vp_y, double vp_w, double vp_h,
                                       double lw, uint8_t* fb) {
  if (!fb || w <= 0 || h <= 0) {
    sf_set_error("sf_image_explosion_host: bad argument");
    return SF_ERR_ARG;
  }
  const sfh::Geometry g{w, h, (double)w / vp_w, (double)h / vp_h, vp_x, vp_y, lw};
  sfh::draw_explosion(x, y, g, fb);
  return SF_OK;
}

// test hook: ONE arc cairo_arc(xc, yc, r, a1, a2) + cairo_stroke (a2 - a1 <= pi/2: one Bezier segment) as 8-bit coverage
extern "C" int sf_image_arc_alpha(double xc, double yc, double r, double a1, double a2, int w, int h, double vp_x, double vp_y,
                                  double vp_w, double vp_h, double lw, uint8_t* alpha) {
  if (!alpha || w <= 0 || h <= 0 || !(a2 > a1) || a2 - a1 > 1.5707963267948966) {
    sf_set_error("sf_image_arc_alpha: bad argument");
    return SF_ERR_ARG;
  }
  const sfh::Geometry g{w, h, (double)w / vp_w, (double)h / vp_h, vp_x, vp_y, lw};
  const sft::Affine v = sft::view_matrix(g.sx, g.sy, g.vp_x, g.vp_y);
  int px[34], py[34], tx[34], ty[34];
  const int n = sft::flatten_faces(sft::arc_knots(v, xc, yc, sft::arc_k(r, a1, a2)), px, py, tx, ty, 33);
  sfh::Object ob;
  ob.chain = true;
  ob.nq = n - 1;
  for (int p = 0; p + 1 < n; p++)
    ob.q[p] = sft::faces_quad(px[p], py[p], tx[p], ty[p], px[p + 1], py[p + 1], tx[p + 1], ty[p + 1], g.sx, g.sy, (double)(float)lw / 2.0);
  ob.nsrc = 0;
  for (int i = 0; i < ob.nq; i++) { ob.src_members[ob.nsrc] = 1u << i; ob.src_sign[ob.nsrc++] = 1; }
  std::vector<int> acc((size_t)w * h, 0);
  sfh::object_coverage(ob, w, h, acc.data());
  for (int i = 0; i < w * h; i++) alpha[i] = (uint8_t)sft::area_to_alpha(acc[i]);
  return ob.nq;
}
