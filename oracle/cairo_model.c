/*
 * cairo_model.c -- TEST INFRASTRUCTURE ONLY (never linked into libsfmi.so; loaded by tests/ and oracle/render_np.py).
 *
 * A plain-C restatement of what cairo 1.16.0's IMAGE backend does with the handful of calls the reference's renderer
 * makes (SRC/draw.cpp:82-270: scale / translate / rotate, move_to / line_to / close_path / arc / rectangle,
 * set_line_width, set_source_rgb, stroke, fill, paint; default antialias, butt caps, miter joins, miter limit 10,
 * tolerance 0.1, operator OVER on a CAIRO_FORMAT_RGB24 surface).
 *
 * cairo is a third-party dependency of the reference (python/spacefortress/setup.py:8 links the system's through
 * pkg-config) and is not under /root/reference; the version in this image is conda's 1.16.0.  What is restated here is
 * its published algorithm (LGPL-2.1 / MPL-1.1 sources, named by file and function below) -- written from the algorithm,
 * not copied -- and PINNED two ways: the same draw scripts through the real library (oracle/cairo_probe.c,
 * tests/test_cairo_model.py: random strokes, fills, arcs, poses) and the reference's own frames drawn by its real
 * draw.cpp (tests/golden/frames/).
 *
 *   user -> device, matrices            cairo-matrix.c  cairo_matrix_multiply / _init_rotate / transform_point,
 *                                       cairo-gstate.c  _cairo_gstate_scale / _translate / _rotate
 *   24.8 fixed point                    cairo-fixed-private.h  _cairo_fixed_from_double (magic-number rounding)
 *   path bookkeeping                    cairo-path-fixed.c  _cairo_path_fixed_move_to / _line_to / _close_path
 *   arcs -> Bezier segments             cairo-arc.c  _cairo_arc_in_direction, _arc_segments_needed, _cairo_arc_segment
 *   Bezier flattening                   cairo-spline.c  _cairo_spline_init / _decompose / _error_squared, _de_casteljau
 *   stroke -> polygon                   cairo-path-stroke-polygon.c  compute_face, line_to, curve_to / spline_to,
 *                                       inner_join, outer_join (miter), add_caps (butt), close_path
 *   rectilinear stroke -> boxes         cairo-path-stroke-boxes.c  _cairo_rectilinear_stroker_emit_segments
 *   polygon edges, clipping to limits   cairo-polygon.c  _cairo_polygon_add_edge, _add_clipped_edge
 *   polygon -> coverage                 cairo-tor-scan-converter.c  polygon_add_edge, sub_row, full_row,
 *                                       cell_list_render_edge, can_do_full_row, blit (GRID 256 x 15)
 *   boxes -> coverage                   cairo-rectangular-scan-converter.c  _active_edges_to_spans
 *   coverage -> pixels                  cairo-image-compositor.c  _fill_xrgb32_lerp_opaque_spans, lerp8x4 / mul8x2_8
 *   which of these a call reaches       cairo-spans-compositor.c  _cairo_spans_compositor_stroke / _fill
 */
#include "cairo_model.h"

#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stdlib.h>
#include <string.h>

typedef struct { double xx, yx, xy, yy, x0, y0; } mat_t;
typedef struct { int32_t x, y; } pt_t;
typedef struct { pt_t p1, p2; int32_t top, bottom; int dir; } edge_t;

#define FRAC_BITS 8
#define FIXED_ONE 256

/* ---- fixed point --------------------------------------------------------------------------------------------- */
static int32_t fx_from_double(double d) { /* round to nearest, ties to even, at 2^-8 (the 1.5 * 2^44 trick) */
  union { double d; int32_t i[2]; } u;
  u.d = d + 26388279066624.0; /* (1LL << (52 - 8)) * 1.5 */
  return u.i[0];
}
static double fx_to_double(int32_t f) { return (double)f / 256.0; }
static int32_t fx_floor_int(int32_t f) { return f >> 8; }
static int32_t fx_ceil_int(int32_t f) { return (f + 255) >> 8; }

static int32_t mul_div_floor(int32_t a, int32_t b, int32_t c) {
  /* _cairo_fixed_mul_div_floor = _cairo_int64_32_div = a plain C division: it truncates towards zero, whatever its name
   * (seen from outside: an edge leaving the surface to the right is cut one 1/256 lower than a true floor would) */
  return (int32_t)(((int64_t)a * b) / c);
}

/* ---- matrices ------------------------------------------------------------------------------------------------ */
static void mat_mul(mat_t* r, const mat_t* a, const mat_t* b) {
  mat_t t;
  t.xx = a->xx * b->xx + a->yx * b->xy;
  t.yx = a->xx * b->yx + a->yx * b->yy;
  t.xy = a->xy * b->xx + a->yy * b->xy;
  t.yy = a->xy * b->yx + a->yy * b->yy;
  t.x0 = a->x0 * b->xx + a->y0 * b->xy + b->x0;
  t.y0 = a->x0 * b->yx + a->y0 * b->yy + b->y0;
  *r = t;
}
static void mat_init(mat_t* m, double xx, double yx, double xy, double yy, double x0, double y0) {
  m->xx = xx; m->yx = yx; m->xy = xy; m->yy = yy; m->x0 = x0; m->y0 = y0;
}
static void mat_distance(const mat_t* m, double* dx, double* dy) {
  double nx = m->xx * *dx + m->xy * *dy;
  double ny = m->yx * *dx + m->yy * *dy;
  *dx = nx; *dy = ny;
}
static void mat_point(const mat_t* m, double* x, double* y) {
  mat_distance(m, x, y);
  *x += m->x0; *y += m->y0;
}
static int mat_is_identity(const mat_t* m) {
  return m->xx == 1.0 && m->yx == 0.0 && m->xy == 0.0 && m->yy == 1.0 && m->x0 == 0.0 && m->y0 == 0.0;
}

/* ---- the drawing state --------------------------------------------------------------------------------------- */
enum { OP_MOVE, OP_LINE, OP_CURVE, OP_CLOSE };
typedef struct { int op; pt_t p[3]; } pop_t;

typedef struct {
  mat_t ctm, inv;
  double lw;
  int grey;
} gs_t;

typedef struct {
  int w, h;
  uint8_t* fb;
  gs_t gs, stack[16];
  int sp;
  pop_t* ops;
  int nops, cap;
  int has_current, needs_move, has_last_move;
  pt_t cur, last_move;
  int rectilinear, has_curve;
} ctx_t;

static edge_t* g_edges;
static int g_nedges, g_ecap;

static void poly_reset(void) { g_nedges = 0; }
static void poly_push(const pt_t* p1, const pt_t* p2, int top, int bottom, int dir) {
  if (top >= bottom) return;
  if (g_nedges == g_ecap) {
    g_ecap = g_ecap ? 2 * g_ecap : 256;
    g_edges = (edge_t*)realloc(g_edges, sizeof(edge_t) * g_ecap);
  }
  edge_t* e = &g_edges[g_nedges++];
  e->p1 = *p1; e->p2 = *p2; e->top = top; e->bottom = bottom; e->dir = dir;
}

/* x of the line p1-p2 at y / y at x: cairo-polygon.c? no: cairo-line.c / cairo-misc: _cairo_edge_compute_intersection_* */
static int32_t edge_x_for_y(const pt_t* p1, const pt_t* p2, int32_t y) {
  if (y == p1->y) return p1->x;
  if (y == p2->y) return p2->x;
  int32_t x = p1->x, dy = p2->y - p1->y;
  if (dy != 0) x += mul_div_floor(y - p1->y, p2->x - p1->x, dy);
  return x;
}
static int32_t edge_y_for_x(const pt_t* p1, const pt_t* p2, int32_t x) {
  if (x == p1->x) return p1->y;
  if (x == p2->x) return p2->y;
  int32_t y = p1->y, dx = p2->x - p1->x;
  if (dx != 0) y += mul_div_floor(x - p1->x, p2->y - p1->y, dx);
  return y;
}

/* cairo-polygon.c: _add_clipped_edge against ONE limit box (the surface) */
static void poly_add_clipped(const pt_t* p1, const pt_t* p2, int top, int bottom, int dir, const pt_t* l1,
                             const pt_t* l2) {
  pt_t bot_left = {l1->x, l2->y}, top_right = {l2->x, l1->y};
  if (top >= l2->y || bottom <= l1->y) return;
  int32_t top_y = top > l1->y ? top : l1->y;
  int32_t bot_y = bottom < l2->y ? bottom : l2->y;
  int32_t pleft = p1->x < p2->x ? p1->x : p2->x;
  int32_t pright = p1->x > p2->x ? p1->x : p2->x;
  if (l1->x <= pleft && pright <= l2->x) {
    poly_push(p1, p2, top_y, bot_y, dir);
  } else if (pright <= l1->x) {
    poly_push(l1, &bot_left, top_y, bot_y, dir);
  } else if (l2->x <= pleft) {
    poly_push(&top_right, l2, top_y, bot_y, dir);
  } else {
    int32_t left_y, right_y;
    int tl_br = (p1->x <= p2->x) == (p1->y <= p2->y);
    if (tl_br) {
      if (pleft >= l1->x) {
        left_y = top_y;
      } else {
        left_y = edge_y_for_x(p1, p2, l1->x);
        if (edge_x_for_y(p1, p2, left_y) < l1->x) left_y++;
      }
      if (left_y > bot_y) left_y = bot_y;
      if (top_y < left_y) {
        poly_push(l1, &bot_left, top_y, left_y, dir);
        top_y = left_y;
      }
      if (pright <= l2->x) {
        right_y = bot_y;
      } else {
        right_y = edge_y_for_x(p1, p2, l2->x);
        if (edge_x_for_y(p1, p2, right_y) > l2->x) right_y--;
      }
      if (right_y < top_y) right_y = top_y;
      if (bot_y > right_y) {
        poly_push(&top_right, l2, right_y, bot_y, dir);
        bot_y = right_y;
      }
    } else {
      if (pright <= l2->x) {
        right_y = top_y;
      } else {
        right_y = edge_y_for_x(p1, p2, l2->x);
        if (edge_x_for_y(p1, p2, right_y) > l2->x) right_y++;
      }
      if (right_y > bot_y) right_y = bot_y;
      if (top_y < right_y) {
        poly_push(&top_right, l2, top_y, right_y, dir);
        top_y = right_y;
      }
      if (pleft >= l1->x) {
        left_y = bot_y;
      } else {
        left_y = edge_y_for_x(p1, p2, l1->x);
        if (edge_x_for_y(p1, p2, left_y) < l1->x) left_y--;
      }
      if (left_y < top_y) left_y = top_y;
      if (bot_y > left_y) {
        poly_push(l1, &bot_left, left_y, bot_y, dir);
        bot_y = left_y;
      }
    }
    if (top_y != bot_y) poly_push(p1, p2, top_y, bot_y, dir);
  }
}

static int g_limits_on;
static pt_t g_l1, g_l2;

/* cairo-polygon.c: _cairo_polygon_add_edge */
static void poly_add_edge(const pt_t* a, const pt_t* b, int dir) {
  if (a->y == b->y) return;
  if (a->y > b->y) { const pt_t* t = a; a = b; b = t; dir = -dir; }
  if (g_limits_on) {
    if (b->y <= g_l1.y || a->y >= g_l2.y) return;
    poly_add_clipped(a, b, a->y, b->y, dir, &g_l1, &g_l2);
  } else {
    poly_push(a, b, a->y, b->y, dir);
  }
}

/* ---- coverage -> pixel (cairo-image-compositor.c: mul8x2_8 / add8x2_8x2 / lerp8x4 on one channel) ------------- */
static inline unsigned mul8(unsigned a, unsigned b) {
  unsigned t = a * b + 0x7f;
  return ((t + (t >> 8)) >> 8) & 0xff;
}
static inline unsigned lerp8(unsigned src, unsigned a, unsigned dst) {
  unsigned t = mul8(src, a) + mul8(dst, (~a) & 0xff);
  if (t > 255) t = 255; /* add8x2_8x2 saturates */
  return t;
}
static void put_cov(ctx_t* c, int x, int y, unsigned a, int grey) {
  if (a == 0 || x < 0 || y < 0 || x >= c->w || y >= c->h) return;
  uint8_t* d = &c->fb[y * c->w + x];
  *d = a == 255 ? (uint8_t)grey : (uint8_t)lerp8((unsigned)grey, a, *d);
}

/* ---- cairo-tor-scan-converter.c ------------------------------------------------------------------------------ */
#define GRID_X 256
#define GRID_Y 15
#define AREA_TO_ALPHA(c) (((c) + ((c) << 4) + 256) >> 9) /* GRID_XY = 2 * 256 * 15 */

typedef struct { int32_t quo; int64_t rem; } quorem_t;
typedef struct tedge {
  struct tedge *next, *prev;
  int ytop, height_left, dir, cell;
  quorem_t x, dxdy, dxdy_full;
  int64_t dy;
  int seq;
} tedge_t;

typedef struct { int covered_height; int uncovered_area; } cell_t;

typedef struct {
  int xmin, xmax; /* pixel columns [xmin, xmax) */
  cell_t* cells;  /* index 0 = everything left of xmin, then one per column, then one catch-all on the right */
} cells_t;

static inline cell_t* cell_at(cells_t* cl, int ix) {
  if (ix < cl->xmin) return &cl->cells[0];
  if (ix >= cl->xmax) return &cl->cells[cl->xmax - cl->xmin + 1];
  return &cl->cells[ix - cl->xmin + 1];
}

static inline int to_grid_y(int32_t in) { /* INPUT_TO_GRID_Y: round to the nearest sub-row boundary */
  int64_t t = (int64_t)GRID_Y * in;
  t += 1 << (FRAC_BITS - 1);
  return (int)(t >> FRAC_BITS);
}

static void edge_step(tedge_t* e) {
  if (e->dy == 0) return; /* vertical */
  e->x.quo += e->dxdy.quo;
  e->x.rem += e->dxdy.rem;
  if (e->x.rem < 0) { e->x.quo--; e->x.rem += e->dy; }
  else if (e->x.rem >= e->dy) { e->x.quo++; e->x.rem -= e->dy; }
  e->cell = e->x.quo + (e->x.rem >= e->dy / 2);
}
static void edge_full_step(tedge_t* e) {
  if (e->dy == 0) return;
  e->x.quo += e->dxdy_full.quo;
  e->x.rem += e->dxdy_full.rem;
  if (e->x.rem < 0) { e->x.quo--; e->x.rem += e->dy; }
  else if (e->x.rem >= e->dy) { e->x.quo++; e->x.rem -= e->dy; }
  e->cell = e->x.quo + (e->x.rem >= e->dy / 2);
}

/* polygon_add_edge: the edge in grid units; x sampled at the CENTRE of its first sub-row */
static int tor_init_edge(tedge_t* e, const edge_t* in, int ymin, int ymax) {
  int ytop = to_grid_y(in->top);
  if (ytop < ymin) ytop = ymin;
  int ybot = to_grid_y(in->bottom);
  if (ybot > ymax) ybot = ymax;
  if (ybot <= ytop) return 0;
  e->ytop = ytop;
  e->height_left = ybot - ytop;
  const pt_t *p1, *p2;
  if (in->p2.y > in->p1.y) { e->dir = in->dir; p1 = &in->p1; p2 = &in->p2; }
  else { e->dir = -in->dir; p1 = &in->p2; p2 = &in->p1; }
  if (p2->x == p1->x) {
    e->cell = p1->x;
    e->x.quo = p1->x; e->x.rem = 0;
    e->dxdy.quo = 0; e->dxdy.rem = 0;
    e->dxdy_full.quo = 0; e->dxdy_full.rem = 0;
    e->dy = 0;
  } else {
    int64_t Ex = (int64_t)(p2->x - p1->x) * GRID_X;
    int64_t Ey = (int64_t)(p2->y - p1->y) * GRID_Y * (2 << 8);
    e->dxdy.quo = (int32_t)(Ex * (2 << 8) / Ey);
    e->dxdy.rem = Ex * (2 << 8) % Ey;
    int64_t tmp = (int64_t)(2 * ytop + 1) << 8;
    tmp -= (int64_t)p1->y * GRID_Y * 2;
    tmp *= Ex;
    e->x.quo = (int32_t)(tmp / Ey);
    e->x.rem = tmp % Ey;
    e->x.quo += p1->x;
    if (e->x.rem < 0) { e->x.quo--; e->x.rem += Ey; }
    else if (e->x.rem >= Ey) { e->x.quo++; e->x.rem -= Ey; }
    if (e->height_left >= GRID_Y) {
      tmp = Ex * (2 * GRID_Y << 8);
      e->dxdy_full.quo = (int32_t)(tmp / Ey);
      e->dxdy_full.rem = tmp % Ey;
    } else {
      e->dxdy_full.quo = 0; e->dxdy_full.rem = 0;
    }
    e->cell = e->x.quo + (e->x.rem >= Ey / 2);
    e->dy = Ey;
  }
  return 1;
}

static void add_subspan(cells_t* cl, int x1, int x2) {
  int ix1 = x1 >> 8, fx1 = x1 & 255, ix2 = x2 >> 8, fx2 = x2 & 255;
  if (ix1 != ix2) {
    cell_t* c1 = cell_at(cl, ix1);
    c1->uncovered_area += 2 * fx1;
    c1->covered_height += 1;
    cell_t* c2 = cell_at(cl, ix2);
    c2->uncovered_area -= 2 * fx2;
    c2->covered_height -= 1;
  } else {
    cell_at(cl, ix1)->uncovered_area += 2 * (fx1 - fx2);
  }
}

/* cell_list_render_edge: the exact area one edge adds over a whole pixel row */
static void render_edge(cells_t* cl, tedge_t* e, int sign) {
  quorem_t x1 = e->x;
  edge_full_step(e);
  quorem_t x2 = e->x;
  if (e->dy) { /* step back from the sample location (half a sub-row) to the row's top / bottom */
    x1.quo -= e->dxdy.quo / 2;
    x1.rem -= e->dxdy.rem / 2;
    if (x1.rem < 0) { --x1.quo; x1.rem += e->dy; }
    else if (x1.rem >= e->dy) { ++x1.quo; x1.rem -= e->dy; }
    x2.quo -= e->dxdy.quo / 2;
    x2.rem -= e->dxdy.rem / 2;
    if (x2.rem < 0) { --x2.quo; x2.rem += e->dy; }
    else if (x2.rem >= e->dy) { ++x2.quo; x2.rem -= e->dy; }
  }
  int ix1 = x1.quo >> 8, fx1 = x1.quo & 255, ix2 = x2.quo >> 8, fx2 = x2.quo & 255;
  if (ix1 == ix2) {
    cell_t* c = cell_at(cl, ix1);
    c->covered_height += sign * GRID_Y;
    c->uncovered_area += sign * (fx1 + fx2) * GRID_Y;
    return;
  }
  if (ix2 < ix1) {
    quorem_t tx = x1; x1 = x2; x2 = tx;
    int t = ix1; ix1 = ix2; ix2 = t;
    t = fx1; fx1 = fx2; fx2 = t;
  }
  {
    quorem_t y;
    int64_t tmp, dx;
    int y_last;
    dx = (int64_t)(x2.quo - x1.quo) * e->dy + (x2.rem - x1.rem);
    tmp = (int64_t)(ix1 + 1) * GRID_X * e->dy;
    tmp -= (int64_t)x1.quo * e->dy + x1.rem;
    tmp *= GRID_Y;
    y.quo = (int32_t)(tmp / dx);
    y.rem = tmp % dx;
    cell_t* c = cell_at(cl, ix1);
    c->uncovered_area += sign * y.quo * (GRID_X + fx1);
    c->covered_height += sign * y.quo;
    y_last = y.quo;
    if (ix1 + 1 < ix2) {
      quorem_t dydx_full;
      dydx_full.quo = (int32_t)((int64_t)GRID_Y * GRID_X * e->dy / dx);
      dydx_full.rem = (int64_t)GRID_Y * GRID_X * e->dy % dx;
      ++ix1;
      do {
        y.quo += dydx_full.quo;
        y.rem += dydx_full.rem;
        if (y.rem >= dx) { y.quo++; y.rem -= dx; }
        c = cell_at(cl, ix1);
        c->uncovered_area += sign * (y.quo - y_last) * GRID_X;
        c->covered_height += sign * (y.quo - y_last);
        y_last = y.quo;
        ++ix1;
      } while (ix1 != ix2);
    }
    c = cell_at(cl, ix2);
    c->uncovered_area += sign * (GRID_Y - y_last) * fx2;
    c->covered_height += sign * (GRID_Y - y_last);
  }
}

typedef struct {
  tedge_t head, tail;
  int min_height, is_vertical;
} active_t;

static void active_init(active_t* a) {
  memset(a, 0, sizeof(*a));
  a->head.cell = INT32_MIN; a->head.next = &a->tail; a->head.prev = NULL; a->head.height_left = INT32_MAX;
  a->tail.cell = INT32_MAX; a->tail.prev = &a->head; a->tail.next = NULL; a->tail.height_left = INT32_MAX;
  a->min_height = INT32_MAX;
  a->is_vertical = 1;
}

/* active_list_merge_edges_from_bucket: sort the newcomers by cell, then merge (ties: list elements first) */
static void active_merge(active_t* a, tedge_t** news, int n) {
  for (int i = 1; i < n; i++) { /* stable insertion sort by cell */
    tedge_t* k = news[i];
    int j = i - 1;
    while (j >= 0 && news[j]->cell > k->cell) { news[j + 1] = news[j]; j--; }
    news[j + 1] = k;
  }
  tedge_t* pos = a->head.next;
  for (int i = 0; i < n; i++) {
    tedge_t* e = news[i];
    while (pos != &a->tail && pos->cell <= e->cell) pos = pos->next;
    e->prev = pos->prev; e->next = pos;
    pos->prev->next = e; pos->prev = e;
  }
}

static int can_do_full_row(active_t* a) {
  if (a->min_height <= 0) {
    int mh = INT32_MAX, vert = 1;
    for (tedge_t* e = a->head.next; e != &a->tail; e = e->next) {
      if (e->height_left < mh) mh = e->height_left;
      vert &= e->dy == 0;
    }
    a->is_vertical = vert;
    a->min_height = mh;
  }
  if (a->min_height < GRID_Y) return 0;
  int prev_x = INT32_MIN;
  for (tedge_t* e = a->head.next; e != &a->tail; e = e->next) {
    int cell;
    if (e->dy) {
      quorem_t x = e->x;
      x.quo += e->dxdy_full.quo;
      x.rem += e->dxdy_full.rem;
      if (x.rem < 0) { x.quo--; x.rem += e->dy; }
      else if (x.rem >= e->dy) { x.quo++; x.rem -= e->dy; }
      cell = x.quo + (x.rem >= e->dy / 2);
    } else {
      cell = e->cell;
    }
    if (cell < prev_x) return 0;
    prev_x = cell;
  }
  return 1;
}

static void dec_edge(active_t* a, tedge_t* e, int h) {
  e->height_left -= h;
  if (e->height_left == 0) {
    e->prev->next = e->next;
    e->next->prev = e->prev;
    a->min_height = -1;
  }
}

static void full_row(active_t* a, cells_t* cl) {
  tedge_t* left = a->head.next;
  while (left != &a->tail) {
    tedge_t* right;
    int winding;
    dec_edge(a, left, GRID_Y);
    winding = left->dir;
    right = left->next;
    do {
      dec_edge(a, right, GRID_Y);
      winding += right->dir;
      if (winding == 0 && right->next->cell != right->cell) break;
      edge_full_step(right);
      right = right->next;
    } while (1);
    render_edge(cl, left, +1);
    render_edge(cl, right, -1);
    left = right->next;
  }
}

static void sub_row(active_t* a, cells_t* cl) {
  tedge_t* e = a->head.next;
  int xstart = INT32_MIN, prev_x = INT32_MIN, winding = 0;
  while (e != &a->tail) {
    tedge_t* next = e->next;
    int xend = e->cell;
    if (--e->height_left) {
      edge_step(e);
      if (e->cell < prev_x) {
        tedge_t* pos = e->prev;
        pos->next = next;
        next->prev = pos;
        do { pos = pos->prev; } while (e->cell < pos->cell);
        pos->next->prev = e;
        e->next = pos->next;
        e->prev = pos;
        pos->next = e;
      } else {
        prev_x = e->cell;
      }
      a->min_height = -1;
    } else {
      e->prev->next = next;
      next->prev = e->prev;
      a->min_height = -1;
    }
    winding += e->dir;
    if (winding == 0) {
      if (next->cell != xend) {
        add_subspan(cl, xstart, xend);
        xstart = INT32_MIN;
      }
    } else if (xstart == INT32_MIN) {
      xstart = xend;
    }
    e = next;
  }
}

/* glitter_scan_converter_render + blit: the polygon in g_edges, winding rule, onto the frame in `grey` */
static void tor_render(ctx_t* c, int grey) {
  if (g_nedges == 0) return;
  /* the window: polygon extents (as _add_edge tracks them) rounded out to pixels, inside the surface */
  int32_t ex1 = INT32_MAX, ey1 = INT32_MAX, ex2 = INT32_MIN, ey2 = INT32_MIN;
  for (int i = 0; i < g_nedges; i++) {
    const edge_t* e = &g_edges[i];
    if (e->top < ey1) ey1 = e->top;
    if (e->bottom > ey2) ey2 = e->bottom;
    int32_t x = e->top == e->p1.y ? e->p1.x : edge_x_for_y(&e->p1, &e->p2, e->top);
    if (x < ex1) ex1 = x;
    if (x > ex2) ex2 = x;
    x = e->bottom == e->p2.y ? e->p2.x : edge_x_for_y(&e->p1, &e->p2, e->bottom);
    if (x < ex1) ex1 = x;
    if (x > ex2) ex2 = x;
  }
  int xmin = fx_floor_int(ex1), ymin = fx_floor_int(ey1), xmax = fx_ceil_int(ex2), ymax = fx_ceil_int(ey2);
  if (xmin < 0) xmin = 0;
  if (ymin < 0) ymin = 0;
  if (xmax > c->w) xmax = c->w;
  if (ymax > c->h) ymax = c->h;
  if (xmin >= xmax || ymin >= ymax) return;

  int h = ymax - ymin;
  tedge_t* te = (tedge_t*)calloc((size_t)g_nedges, sizeof(tedge_t));
  int nte = 0;
  for (int i = 0; i < g_nedges; i++)
    if (tor_init_edge(&te[nte], &g_edges[i], ymin * GRID_Y, ymax * GRID_Y)) { te[nte].seq = nte; nte++; }
  cells_t cl;
  cl.xmin = xmin; cl.xmax = xmax;
  int ncells = xmax - xmin + 2;
  cl.cells = (cell_t*)calloc((size_t)ncells, sizeof(cell_t));
  active_t act;
  active_init(&act);
  tedge_t** news = (tedge_t**)malloc(sizeof(tedge_t*) * (size_t)(nte + 1));
  /* has this pixel row any starting edge? */
  unsigned char* row_has = (unsigned char*)calloc((size_t)h + 1, 1);
  for (int k = 0; k < nte; k++) row_has[te[k].ytop / GRID_Y - ymin] = 1;

  for (int i = 0, j; i < h; i = j) {
    int do_full = 0;
    j = i + 1;
    int rowy = (i + ymin) * GRID_Y;
    /* polygon_fill_buckets */
    int max_suby = 0, nrow = 0;
    if (row_has[i]) {
      for (int k = 0; k < nte; k++) {
        tedge_t* e = &te[k];
        if (e->ytop >= rowy && e->ytop < rowy + GRID_Y) {
          int suby = e->ytop - rowy;
          if (suby > max_suby) max_suby = suby;
          if (e->height_left < act.min_height) act.min_height = e->height_left;
          act.is_vertical &= e->dy == 0;
          nrow++;
        }
      }
    }
    if (max_suby == 0) {
      if (nrow) {
        int n = 0;
        for (int k = 0; k < nte; k++)
          if (te[k].ytop == rowy) news[n++] = &te[k];
        active_merge(&act, news, n);
      }
      if (act.head.next == &act.tail) {
        act.min_height = INT32_MAX;
        act.is_vertical = 1;
        for (; j < h && !row_has[j]; j++) {}
        continue;
      }
      do_full = can_do_full_row(&act);
    }
    if (do_full) {
      full_row(&act, &cl);
      if (act.is_vertical) {
        while (j < h && !row_has[j] && act.min_height >= 2 * GRID_Y) {
          act.min_height -= GRID_Y;
          j++;
        }
        if (j != i + 1) {
          int count = j - (i + 1);
          for (tedge_t* e = act.head.next; e != &act.tail;) { /* step_edges */
            tedge_t* nx = e->next;
            e->height_left -= GRID_Y * count;
            if (!e->height_left) { e->prev->next = e->next; e->next->prev = e->prev; act.min_height = -1; }
            e = nx;
          }
        }
      }
    } else {
      for (int sub = 0; sub < GRID_Y; sub++) {
        if (row_has[i] && !(max_suby == 0 && sub == 0)) { /* (sub-row 0's edges of a max_suby == 0 row are in already) */
          int n = 0;
          for (int k = 0; k < nte; k++)
            if (te[k].ytop == rowy + sub) news[n++] = &te[k];
          if (n) active_merge(&act, news, n);
        }
        sub_row(&act, &cl);
      }
    }
    /* blit */
    {
      int cover = cl.cells[0].covered_height * GRID_X * 2;
      for (int x = xmin; x < xmax; x++) {
        cell_t* ce = &cl.cells[x - xmin + 1];
        cover += ce->covered_height * GRID_X * 2;
        int area = cover - ce->uncovered_area;
        unsigned a = (unsigned)AREA_TO_ALPHA(area);
        if (a > 255) a = 255;
        for (int yy = i; yy < j; yy++) put_cov(c, x, yy + ymin, a, grey);
      }
    }
    memset(cl.cells, 0, sizeof(cell_t) * (size_t)ncells);
    act.min_height -= GRID_Y;
  }
  free(row_has); free(news); free(cl.cells); free(te);
}

/* ---- boxes: cairo-rectangular-scan-converter.c ---------------------------------------------------------------- */
typedef struct { int32_t x1, y1, x2, y2; } box_t;

/* coverage of the UNION of axis-aligned boxes (the tessellation removes overlaps first:
 * _cairo_bentley_ottmann_tessellate_boxes in _cairo_path_fixed_stroke_rectilinear_to_boxes), exact area per pixel in
 * 1/65536, c = area >> 8, alpha = c - (c >> 8)  (_active_edges_to_spans) */
static void boxes_render(ctx_t* c, const box_t* bx, int nb, int grey) {
  /* compress coordinates: the union's area inside a pixel = sum over the grid of distinct x / y breaks */
  int32_t xs[64], ys[64];
  if (nb > 16) nb = 16;
  int32_t minx = INT32_MAX, miny = INT32_MAX, maxx = INT32_MIN, maxy = INT32_MIN;
  for (int i = 0; i < nb; i++) {
    if (bx[i].x1 < minx) minx = bx[i].x1;
    if (bx[i].y1 < miny) miny = bx[i].y1;
    if (bx[i].x2 > maxx) maxx = bx[i].x2;
    if (bx[i].y2 > maxy) maxy = bx[i].y2;
  }
  for (int py = fx_floor_int(miny); py < fx_ceil_int(maxy); py++)
    for (int px = fx_floor_int(minx); px < fx_ceil_int(maxx); px++) {
      if (px < 0 || py < 0 || px >= c->w || py >= c->h) continue;
      int nx = 0, ny = 0;
      int32_t X0 = px * 256, X1 = X0 + 256, Y0 = py * 256, Y1 = Y0 + 256;
      xs[nx++] = X0; xs[nx++] = X1; ys[ny++] = Y0; ys[ny++] = Y1;
      for (int i = 0; i < nb; i++) {
        if (bx[i].x1 > X0 && bx[i].x1 < X1) xs[nx++] = bx[i].x1;
        if (bx[i].x2 > X0 && bx[i].x2 < X1) xs[nx++] = bx[i].x2;
        if (bx[i].y1 > Y0 && bx[i].y1 < Y1) ys[ny++] = bx[i].y1;
        if (bx[i].y2 > Y0 && bx[i].y2 < Y1) ys[ny++] = bx[i].y2;
      }
      for (int i = 1; i < nx; i++) { int32_t k = xs[i]; int j = i - 1; while (j >= 0 && xs[j] > k) { xs[j + 1] = xs[j]; j--; } xs[j + 1] = k; }
      for (int i = 1; i < ny; i++) { int32_t k = ys[i]; int j = i - 1; while (j >= 0 && ys[j] > k) { ys[j + 1] = ys[j]; j--; } ys[j + 1] = k; }
      int64_t area = 0;
      for (int a = 0; a + 1 < nx; a++)
        for (int b = 0; b + 1 < ny; b++) {
          if (xs[a] == xs[a + 1] || ys[b] == ys[b + 1]) continue;
          int in = 0;
          for (int i = 0; i < nb && !in; i++)
            in = bx[i].x1 <= xs[a] && xs[a + 1] <= bx[i].x2 && bx[i].y1 <= ys[b] && ys[b + 1] <= bx[i].y2;
          if (in) area += (int64_t)(xs[a + 1] - xs[a]) * (ys[b + 1] - ys[b]);
        }
      int cv = (int)(area >> 8);
      put_cov(c, px, py, (unsigned)(cv - (cv >> 8)), grey);
    }
}

/* ---- path ---------------------------------------------------------------------------------------------------- */
static void path_reset(ctx_t* c) {
  c->nops = 0;
  c->has_current = 0; c->needs_move = 1; c->has_last_move = 0;
  c->rectilinear = 1; c->has_curve = 0;
}
static void path_push(ctx_t* c, int op, const pt_t* p, int np) {
  if (c->nops == c->cap) {
    c->cap = c->cap ? 2 * c->cap : 64;
    c->ops = (pop_t*)realloc(c->ops, sizeof(pop_t) * c->cap);
  }
  c->ops[c->nops].op = op;
  for (int i = 0; i < np; i++) c->ops[c->nops].p[i] = p[i];
  c->nops++;
}
static void path_move_to_fixed(ctx_t* c, pt_t p) { /* _cairo_path_fixed_move_to: the op is added lazily */
  c->needs_move = 1;
  c->has_current = 1;
  c->cur = p;
  c->last_move = p;
}
static void path_apply_move(ctx_t* c) {
  if (!c->needs_move) return;
  c->needs_move = 0;
  path_push(c, OP_MOVE, &c->cur, 1);
}
static int path_last_op(ctx_t* c) { return c->nops ? c->ops[c->nops - 1].op : -1; }
static void path_line_to_fixed(ctx_t* c, pt_t p) { /* _cairo_path_fixed_line_to */
  if (!c->has_current) { path_move_to_fixed(c, p); return; }
  path_apply_move(c);
  if (path_last_op(c) != OP_MOVE) {
    if (p.x == c->cur.x && p.y == c->cur.y) return;
  }
  if (path_last_op(c) == OP_LINE && c->nops >= 2) {
    /* previous point */
    pop_t* prev_op = &c->ops[c->nops - 2];
    pt_t pp = prev_op->op == OP_CURVE ? prev_op->p[2] : prev_op->p[0];
    if (pp.x == c->cur.x && pp.y == c->cur.y) {
      c->nops--; /* previous line element was degenerate */
    } else {
      int64_t adx = c->cur.x - pp.x, ady = c->cur.y - pp.y, bdx = p.x - c->cur.x, bdy = p.y - c->cur.y;
      if (ady * bdx == bdy * adx && !(adx * bdx + ady * bdy < 0)) c->nops--; /* same gradient, not backwards */
    }
  }
  if (c->rectilinear) c->rectilinear = c->cur.x == p.x || c->cur.y == p.y;
  path_push(c, OP_LINE, &p, 1);
  c->cur = p;
}
static void path_curve_to_fixed(ctx_t* c, pt_t b, pt_t cc, pt_t d) {
  if (!c->has_current) path_move_to_fixed(c, b);
  /* (cairo drops a curve whose four points coincide; not reached by the scripts here) */
  path_apply_move(c);
  pt_t p[3] = {b, cc, d};
  path_push(c, OP_CURVE, p, 3);
  c->cur = d;
  c->rectilinear = 0;
  c->has_curve = 1;
}
static void path_close(ctx_t* c) {
  if (!c->has_current) return;
  /* _cairo_path_fixed_close_path: line back to the last move point, then the op */
  path_line_to_fixed(c, c->last_move);
  if (path_last_op(c) == OP_LINE && c->nops >= 1) {
    /* a closing line_to that the close op itself implies is dropped */
    pop_t* l = &c->ops[c->nops - 1];
    if (l->p[0].x == c->last_move.x && l->p[0].y == c->last_move.y) c->nops--;
  }
  c->needs_move = 1;
  path_push(c, OP_CLOSE, &c->last_move, 0);
  c->cur = c->last_move;
}

static pt_t user_to_fixed(ctx_t* c, double x, double y) {
  mat_point(&c->gs.ctm, &x, &y);
  pt_t p = {fx_from_double(x), fx_from_double(y)};
  return p;
}

/* cairo-arc.c */
static double arc_max_angle(double tolerance) {
  static const struct { double angle, error; } table[] = {
      {M_PI / 1.0, 0.0185185185185185036127},   {M_PI / 2.0, 0.000272567143730179811158},
      {M_PI / 3.0, 2.38647043651461047433e-05}, {M_PI / 4.0, 4.2455377443222443279e-06},
      {M_PI / 5.0, 1.11281001494389081528e-06}, {M_PI / 6.0, 3.72662000942734705475e-07},
      {M_PI / 7.0, 1.47783685574284411325e-07}, {M_PI / 8.0, 6.63240432022601149057e-08},
      {M_PI / 9.0, 3.2715520137536980553e-08},  {M_PI / 10.0, 1.73863223499021216974e-08},
      {M_PI / 11.0, 9.81410988043554039085e-09},
  };
  for (int i = 0; i < 11; i++)
    if (table[i].error < tolerance) return table[i].angle;
  return M_PI / 12.0; /* finer than anything the scripts here reach */
}
static double circle_major_axis(const mat_t* m, double radius) { /* _cairo_matrix_transformed_circle_major_axis */
  double a = m->xx, b = m->yx, c = m->xy, d = m->yy;
  double i = a * a + b * b, j = c * c + d * d;
  double f = 0.5 * (i + j), g = 0.5 * (i - j), h = a * c + b * d;
  if (fabs(h) == 0 && fabs(g) == 0) return radius * sqrt(f); /* (has_unity_scale / uniform: same value) */
  return radius * sqrt(f + hypot(g, h));
}
static void arc_segment(ctx_t* c, double xc, double yc, double radius, double A, double B) {
  double r_sin_A = radius * sin(A), r_cos_A = radius * cos(A);
  double r_sin_B = radius * sin(B), r_cos_B = radius * cos(B);
  double h = 4.0 / 3.0 * tan((B - A) / 4.0);
  pt_t p1 = user_to_fixed(c, xc + r_cos_A - h * r_sin_A, yc + r_sin_A + h * r_cos_A);
  pt_t p2 = user_to_fixed(c, xc + r_cos_B + h * r_sin_B, yc + r_sin_B - h * r_cos_B);
  pt_t p3 = user_to_fixed(c, xc + r_cos_B, yc + r_sin_B);
  path_curve_to_fixed(c, p1, p2, p3);
}
static void arc_in_direction(ctx_t* c, double xc, double yc, double radius, double amin, double amax) {
  if (amax - amin > M_PI) {
    double mid = amin + (amax - amin) / 2.0;
    arc_in_direction(c, xc, yc, radius, amin, mid);
    arc_in_direction(c, xc, yc, radius, mid, amax);
  } else if (amax != amin) {
    double major = circle_major_axis(&c->gs.ctm, radius);
    double max_angle = arc_max_angle(0.1 / major);
    int segments = (int)ceil(fabs(amax - amin) / max_angle);
    double step = (amax - amin) / segments;
    segments -= 1;
    for (int i = 0; i < segments; i++, amin += step) arc_segment(c, xc, yc, radius, amin, amin + step);
    arc_segment(c, xc, yc, radius, amin, amax);
  } else {
    path_line_to_fixed(c, user_to_fixed(c, xc + radius * cos(amin), yc + radius * sin(amin)));
  }
}

/* ---- stroker: cairo-path-stroke-polygon.c -------------------------------------------------------------------- */
typedef struct { double x, y; } dvec_t;
typedef struct {
  pt_t ccw, point, cw;
  pt_t dev_vector; /* slope, fixed */
  dvec_t dev_slope, usr_vector;
  double length;
} face_t;

typedef struct { pt_t* p; int n, cap, dir; } contour_t;
static void contour_add(contour_t* c, const pt_t* p) {
  if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 64; c->p = (pt_t*)realloc(c->p, sizeof(pt_t) * c->cap); }
  c->p[c->n++] = *p;
}
static void contour_to_polygon(contour_t* c) { /* _cairo_polygon_add_contour: an OPEN chain of edges */
  if (c->n <= 1) return;
  for (int i = 1; i < c->n; i++) poly_add_edge(&c->p[i - 1], &c->p[i], c->dir);
}

typedef struct {
  const mat_t *ctm, *inv;
  double half_lw, tolerance, cusp_tolerance;
  int det_positive;
  contour_t cw, ccw;
  pt_t first_point;
  int has_initial_sub_path, has_current_face, has_first_face;
  face_t current_face, first_face;
} stroker_t;

static double normalize_slope(double* dx, double* dy) {
  double dx0 = *dx, dy0 = *dy, mag;
  if (dx0 == 0.0) {
    *dx = 0.0;
    if (dy0 > 0.0) { mag = dy0; *dy = 1.0; } else { mag = -dy0; *dy = -1.0; }
  } else if (dy0 == 0.0) {
    *dy = 0.0;
    if (dx0 > 0.0) { mag = dx0; *dx = 1.0; } else { mag = -dx0; *dx = -1.0; }
  } else {
    mag = hypot(dx0, dy0);
    *dx = dx0 / mag;
    *dy = dy0 / mag;
  }
  return mag;
}

static void compute_face(const pt_t* point, const pt_t* dev_slope, stroker_t* s, face_t* f) {
  double face_dx, face_dy;
  double slope_dx = fx_to_double(dev_slope->x), slope_dy = fx_to_double(dev_slope->y);
  f->length = normalize_slope(&slope_dx, &slope_dy);
  f->dev_slope.x = slope_dx;
  f->dev_slope.y = slope_dy;
  if (!mat_is_identity(s->inv)) {
    mat_distance(s->inv, &slope_dx, &slope_dy);
    normalize_slope(&slope_dx, &slope_dy);
    if (s->det_positive) { face_dx = -slope_dy * s->half_lw; face_dy = slope_dx * s->half_lw; }
    else { face_dx = slope_dy * s->half_lw; face_dy = -slope_dx * s->half_lw; }
    mat_distance(s->ctm, &face_dx, &face_dy);
  } else {
    face_dx = -slope_dy * s->half_lw;
    face_dy = slope_dx * s->half_lw;
  }
  pt_t off = {fx_from_double(face_dx), fx_from_double(face_dy)};
  f->ccw.x = point->x + off.x; f->ccw.y = point->y + off.y;
  f->point = *point;
  f->cw.x = point->x - off.x; f->cw.y = point->y - off.y;
  f->usr_vector.x = slope_dx; f->usr_vector.y = slope_dy;
  f->dev_vector = *dev_slope;
}

static int slope_compare(const pt_t* a, const pt_t* b) { /* _cairo_slope_compare */
  int64_t adx_bdy = (int64_t)a->x * b->y, bdx_ady = (int64_t)b->x * a->y;
  if (a->x == 0 && b->x == 0) return 0;
  if (a->x == 0) return 1;
  if (b->x == 0) return -1;
  if (adx_bdy > bdx_ady) return 1;
  if (adx_bdy < bdx_ady) return -1;
  return 0;
}
/* the full _cairo_slope_compare orders by angle incl. the vertical / anti-parallel special cases; the joins here only
 * need its sign for two non-parallel vectors, which is the sign of the cross product: */
static int join_clockwise_sign(const face_t* in, const face_t* out) {
  int64_t cr = (int64_t)in->dev_vector.x * out->dev_vector.y - (int64_t)out->dev_vector.x * in->dev_vector.y;
  (void)slope_compare;
  return cr > 0 ? 1 : cr < 0 ? -1 : 0; /* > 0: _cairo_slope_compare(in, out) > 0 */
}

static int sgn_cmp(double dx1, double dy1, double dx2, double dy2) {
  double cc = dx1 * dy2 - dx2 * dy1;
  return cc > 0 ? 1 : cc < 0 ? -1 : 0;
}

static void inner_join(stroker_t* s, const face_t* in, const face_t* out, int clockwise) {
  contour_t* inner = clockwise ? &s->ccw : &s->cw;
  contour_add(inner, &in->point);
  contour_add(inner, clockwise ? &out->ccw : &out->cw);
}

static void outer_join(stroker_t* s, const face_t* in, const face_t* out, int clockwise) {
  const pt_t *inpt, *outpt;
  contour_t* outer;
  if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x && in->ccw.y == out->ccw.y) return;
  if (clockwise) { inpt = &in->cw; outpt = &out->cw; outer = &s->cw; }
  else { inpt = &in->ccw; outpt = &out->ccw; outer = &s->ccw; }
  /* miter, limit 10 */
  double in_dot_out = in->dev_slope.x * out->dev_slope.x + in->dev_slope.y * out->dev_slope.y;
  double ml = 10.0;
  if (2 <= ml * ml * (1 + in_dot_out)) {
    double x1 = fx_to_double(inpt->x), y1 = fx_to_double(inpt->y), dx1 = in->dev_slope.x, dy1 = in->dev_slope.y;
    double x2 = fx_to_double(outpt->x), y2 = fx_to_double(outpt->y), dx2 = out->dev_slope.x, dy2 = out->dev_slope.y;
    double my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
    double mx;
    if (fabs(dy1) >= fabs(dy2)) mx = (my - y1) * dx1 / dy1 + x1;
    else mx = (my - y2) * dx2 / dy2 + x2;
    double ix = fx_to_double(in->point.x), iy = fx_to_double(in->point.y);
    double fdx1 = x1 - ix, fdy1 = y1 - iy, fdx2 = x2 - ix, fdy2 = y2 - iy, mdx = mx - ix, mdy = my - iy;
    if (sgn_cmp(fdx1, fdy1, mdx, mdy) != sgn_cmp(fdx2, fdy2, mdx, mdy)) {
      pt_t p = {fx_from_double(mx), fx_from_double(my)};
      outer->p[outer->n - 1] = p;
      outer->p[0] = p; /* (what 1.16 does; right for the closing join, and every contour here is closed) */
      return;
    }
  }
  contour_add(outer, outpt);
}

static void add_caps(stroker_t* s) { /* butt caps */
  if (s->has_current_face) contour_add(&s->ccw, &s->current_face.cw); /* add_trailing_cap */
  contour_to_polygon(&s->ccw);
  s->ccw.n = 0;
  if (s->has_first_face) {
    contour_add(&s->ccw, &s->first_face.cw);
    contour_add(&s->ccw, &s->first_face.ccw); /* add_leading_cap (reversed face: its cw is our ccw) */
    contour_to_polygon(&s->ccw);
    s->ccw.n = 0;
  }
  contour_to_polygon(&s->cw);
  s->cw.n = 0;
}

static void st_move_to(stroker_t* s, const pt_t* p) {
  add_caps(s);
  s->has_first_face = 0;
  s->has_current_face = 0;
  s->has_initial_sub_path = 0;
  s->first_point = *p;
  s->current_face.point = *p;
}

static void st_line_to(stroker_t* s, const pt_t* point) {
  face_t start;
  pt_t* p1 = &s->current_face.point;
  s->has_initial_sub_path = 1;
  if (p1->x == point->x && p1->y == point->y) return;
  pt_t slope = {point->x - p1->x, point->y - p1->y};
  compute_face(p1, &slope, s, &start);
  if (s->has_current_face) {
    int cw = join_clockwise_sign(&s->current_face, &start);
    if (cw) {
      int clockwise = cw > 0; /* = _cairo_slope_compare(in, out) < 0: the turn is towards the ccw side, cw is outside */
      outer_join(s, &s->current_face, &start, clockwise);
      inner_join(s, &s->current_face, &start, clockwise);
    }
  } else {
    if (!s->has_first_face) { s->first_face = start; s->has_first_face = 1; }
    s->has_current_face = 1;
    contour_add(&s->cw, &start.cw);
    contour_add(&s->ccw, &start.ccw);
  }
  s->current_face = start;
  s->current_face.point = *point;
  s->current_face.ccw.x += slope.x; s->current_face.ccw.y += slope.y;
  s->current_face.cw.x += slope.x; s->current_face.cw.y += slope.y;
  contour_add(&s->cw, &s->current_face.cw);
  contour_add(&s->ccw, &s->current_face.ccw);
}

static void st_spline_to(stroker_t* s, const pt_t* point, const pt_t* tangent) {
  face_t face;
  if ((tangent->x | tangent->y) == 0) {
    /* a cusp with a zero tangent: cairo turns the pen around with a fan; not reached by arcs */
    return;
  }
  compute_face(point, tangent, s, &face);
  /* (the round fan at a sharp turn, dot < cusp_tolerance, is not reached: arc pieces turn by <= 22.5 degrees) */
  contour_add(&s->cw, &face.cw);
  contour_add(&s->ccw, &face.ccw);
  s->current_face = face;
}

/* cairo-spline.c */
typedef struct { pt_t a, b, c, d; } knots_t;
typedef struct {
  stroker_t* s; /* stroking */
  int fill;     /* flattening for a fill: points go to the polygon through fill_line_to */
  pt_t last_point;
  knots_t knots;
  pt_t initial_slope, final_slope;
} spline_t;

static void fill_line_to(const pt_t* p);

static void spline_emit(spline_t* sp, const pt_t* point, const pt_t* tangent) {
  if (sp->fill) fill_line_to(point);
  else st_spline_to(sp->s, point, tangent);
}
static void spline_add_point(spline_t* sp, const pt_t* point, const pt_t* knot) {
  if (sp->last_point.x == point->x && sp->last_point.y == point->y) return;
  pt_t slope = {knot->x - point->x, knot->y - point->y};
  sp->last_point = *point;
  spline_emit(sp, point, &slope);
}
static double spline_error_squared(const knots_t* k) {
  double bdx = fx_to_double(k->b.x - k->a.x), bdy = fx_to_double(k->b.y - k->a.y);
  double cdx = fx_to_double(k->c.x - k->a.x), cdy = fx_to_double(k->c.y - k->a.y);
  if (k->a.x != k->d.x || k->a.y != k->d.y) {
    double dx = fx_to_double(k->d.x - k->a.x), dy = fx_to_double(k->d.y - k->a.y);
    double v = dx * dx + dy * dy, u;
    u = bdx * dx + bdy * dy;
    if (u <= 0) {
    } else if (u >= v) { bdx -= dx; bdy -= dy; }
    else { bdx -= u / v * dx; bdy -= u / v * dy; }
    u = cdx * dx + cdy * dy;
    if (u <= 0) {
    } else if (u >= v) { cdx -= dx; cdy -= dy; }
    else { cdx -= u / v * dx; cdy -= u / v * dy; }
  }
  double berr = bdx * bdx + bdy * bdy, cerr = cdx * cdx + cdy * cdy;
  return berr > cerr ? berr : cerr;
}
static void de_casteljau(knots_t* s1, knots_t* s2) {
  pt_t ab = {(s1->a.x + s1->b.x) >> 1, (s1->a.y + s1->b.y) >> 1};
  pt_t bc = {(s1->b.x + s1->c.x) >> 1, (s1->b.y + s1->c.y) >> 1};
  pt_t cd = {(s1->c.x + s1->d.x) >> 1, (s1->c.y + s1->d.y) >> 1};
  pt_t abbc = {(ab.x + bc.x) >> 1, (ab.y + bc.y) >> 1};
  pt_t bccd = {(bc.x + cd.x) >> 1, (bc.y + cd.y) >> 1};
  pt_t fin = {(abbc.x + bccd.x) >> 1, (abbc.y + bccd.y) >> 1};
  s2->a = fin; s2->b = bccd; s2->c = cd; s2->d = s1->d;
  s1->b = ab; s1->c = abbc; s1->d = fin;
}
static void spline_decompose_into(knots_t* s1, double tol2, spline_t* sp) {
  knots_t s2;
  if (spline_error_squared(s1) < tol2) { spline_add_point(sp, &s1->a, &s1->b); return; }
  de_casteljau(s1, &s2);
  spline_decompose_into(s1, tol2, sp);
  spline_decompose_into(&s2, tol2, sp);
}
static int spline_init(spline_t* sp, const pt_t* a, const pt_t* b, const pt_t* c, const pt_t* d) {
  if (a->x == b->x && a->y == b->y && c->x == d->x && c->y == d->y) return 0;
  sp->knots.a = *a; sp->knots.b = *b; sp->knots.c = *c; sp->knots.d = *d;
  const pt_t* q;
  if (a->x != b->x || a->y != b->y) q = b;
  else if (a->x != c->x || a->y != c->y) q = c;
  else if (a->x != d->x || a->y != d->y) q = d;
  else return 0;
  sp->initial_slope.x = q->x - a->x; sp->initial_slope.y = q->y - a->y;
  if (c->x != d->x || c->y != d->y) q = c;
  else if (b->x != d->x || b->y != d->y) q = b;
  else return 0;
  sp->final_slope.x = d->x - q->x; sp->final_slope.y = d->y - q->y;
  return 1;
}
static void spline_decompose(spline_t* sp, double tolerance) {
  knots_t s1 = sp->knots;
  sp->last_point = s1.a;
  spline_decompose_into(&s1, tolerance * tolerance, sp);
  spline_emit(sp, &sp->knots.d, &sp->final_slope);
}

static void st_curve_to(stroker_t* s, const pt_t* b, const pt_t* c, const pt_t* d) {
  spline_t sp;
  face_t face;
  memset(&sp, 0, sizeof(sp));
  sp.s = s;
  if (!spline_init(&sp, &s->current_face.point, b, c, d)) { st_line_to(s, d); return; }
  compute_face(&s->current_face.point, &sp.initial_slope, s, &face);
  if (s->has_current_face) {
    int clockwise = join_clockwise_sign(&s->current_face, &face) > 0;
    outer_join(s, &s->current_face, &face, clockwise);
    inner_join(s, &s->current_face, &face, clockwise);
  } else {
    if (!s->has_first_face) { s->first_face = face; s->has_first_face = 1; }
    s->has_current_face = 1;
    contour_add(&s->cw, &face.cw);
    contour_add(&s->ccw, &face.ccw);
  }
  s->current_face = face;
  s->has_initial_sub_path = 1;
  spline_decompose(&sp, s->tolerance);
}

static void st_close(stroker_t* s) {
  st_line_to(s, &s->first_point);
  if (s->has_first_face && s->has_current_face) {
    int cw = join_clockwise_sign(&s->current_face, &s->first_face);
    int clockwise = cw > 0;
    outer_join(s, &s->current_face, &s->first_face, clockwise);
    inner_join(s, &s->current_face, &s->first_face, clockwise);
    contour_to_polygon(&s->cw);
    contour_to_polygon(&s->ccw);
    s->cw.n = 0; s->ccw.n = 0;
  } else {
    add_caps(s);
  }
  s->has_initial_sub_path = 0;
  s->has_first_face = 0;
  s->has_current_face = 0;
}

static contour_t g_cw, g_ccw;

static void stroke_to_polygon(ctx_t* c) {
  stroker_t s;
  memset(&s, 0, sizeof(s));
  s.ctm = &c->gs.ctm; s.inv = &c->gs.inv;
  s.half_lw = c->gs.lw / 2.0;
  s.tolerance = 0.1;
  s.cusp_tolerance = 1 - s.tolerance / s.half_lw;
  s.cusp_tolerance *= s.cusp_tolerance; s.cusp_tolerance *= 2; s.cusp_tolerance -= 1;
  s.det_positive = (c->gs.ctm.xx * c->gs.ctm.yy - c->gs.ctm.yx * c->gs.ctm.xy) >= 0.0;
  g_cw.n = 0; g_ccw.n = 0;
  s.cw = g_cw; s.cw.dir = 1;
  s.ccw = g_ccw; s.ccw.dir = -1;
  for (int i = 0; i < c->nops; i++) {
    pop_t* o = &c->ops[i];
    switch (o->op) {
      case OP_MOVE: st_move_to(&s, &o->p[0]); break;
      case OP_LINE: st_line_to(&s, &o->p[0]); break;
      case OP_CURVE: st_curve_to(&s, &o->p[0], &o->p[1], &o->p[2]); break;
      case OP_CLOSE: st_close(&s); break;
    }
  }
  add_caps(&s);
  g_cw = s.cw; g_ccw = s.ccw;
}

/* cairo-path-stroke-boxes.c: open butt-capped axis-aligned segments, scale-only matrix */
static int stroke_rectilinear_boxes(ctx_t* c, box_t* out, int cap) {
  const mat_t* m = &c->gs.ctm;
  if (!(m->xy == 0.0 && m->yx == 0.0)) return -1; /* _cairo_matrix_is_scale */
  int32_t hx = fx_from_double(fabs(m->xx) * c->gs.lw / 2.0), hy = fx_from_double(fabs(m->yy) * c->gs.lw / 2.0);
  int n = 0;
  pt_t cur = {0, 0};
  int seg_in_sub = 0;
  for (int i = 0; i < c->nops; i++) {
    pop_t* o = &c->ops[i];
    if (o->op == OP_MOVE) { cur = o->p[0]; seg_in_sub = 0; continue; }
    if (o->op != OP_LINE) return -1; /* closed sub-paths (joins) are not needed by the scripts here */
    if (seg_in_sub) return -1;       /* ... nor poly-lines */
    seg_in_sub = 1;
    pt_t a = cur, b = o->p[0];
    if (a.x == b.x && a.y == b.y) { cur = b; continue; }
    if (a.y == b.y) { a.y -= hy; b.y += hy; } else { a.x -= hx; b.x += hx; }
    if (n == cap) return -1;
    out[n].x1 = a.x < b.x ? a.x : b.x; out[n].x2 = a.x < b.x ? b.x : a.x;
    out[n].y1 = a.y < b.y ? a.y : b.y; out[n].y2 = a.y < b.y ? b.y : a.y;
    n++;
    cur = b;
  }
  return n;
}

/* cairo-path-fill.c: _cairo_path_fixed_fill_to_polygon */
static pt_t g_fill_cur, g_fill_first;
static int g_fill_has;
static void fill_line_to(const pt_t* p) {
  poly_add_edge(&g_fill_cur, p, 1);
  g_fill_cur = *p;
}
static void fill_to_polygon(ctx_t* c) {
  g_fill_has = 0;
  for (int i = 0; i < c->nops; i++) {
    pop_t* o = &c->ops[i];
    switch (o->op) {
      case OP_MOVE:
        if (g_fill_has) fill_line_to(&g_fill_first);
        g_fill_cur = g_fill_first = o->p[0];
        g_fill_has = 1;
        break;
      case OP_LINE: fill_line_to(&o->p[0]); break;
      case OP_CURVE: {
        spline_t sp;
        memset(&sp, 0, sizeof(sp));
        sp.fill = 1;
        if (!spline_init(&sp, &g_fill_cur, &o->p[0], &o->p[1], &o->p[2])) { fill_line_to(&o->p[2]); break; }
        spline_decompose(&sp, 0.1);
        break;
      }
      case OP_CLOSE: if (g_fill_has) fill_line_to(&g_fill_first); break;
    }
  }
  if (g_fill_has) fill_line_to(&g_fill_first);
}

static int path_is_box(ctx_t* c, box_t* b) { /* one move + three lines + close (cairo_rectangle), axis-aligned */
  if (!c->rectilinear) return 0;
  pt_t pts[8];
  int n = 0;
  for (int i = 0; i < c->nops; i++) {
    pop_t* o = &c->ops[i];
    if (o->op == OP_MOVE) { if (n) return 0; pts[n++] = o->p[0]; }
    else if (o->op == OP_LINE) { if (n == 0 || n >= 6) return 0; pts[n++] = o->p[0]; }
    else if (o->op == OP_CLOSE) { if (i != c->nops - 1) return 0; }
    else return 0;
  }
  if (n == 5 && pts[4].x == pts[0].x && pts[4].y == pts[0].y) n = 4;
  if (n != 4) return 0;
  int ok = (pts[0].y == pts[1].y && pts[1].x == pts[2].x && pts[2].y == pts[3].y && pts[3].x == pts[0].x) ||
           (pts[0].x == pts[1].x && pts[1].y == pts[2].y && pts[2].x == pts[3].x && pts[3].y == pts[0].y);
  if (!ok) return 0;
  b->x1 = pts[0].x < pts[2].x ? pts[0].x : pts[2].x; b->x2 = pts[0].x < pts[2].x ? pts[2].x : pts[0].x;
  b->y1 = pts[0].y < pts[2].y ? pts[0].y : pts[2].y; b->y2 = pts[0].y < pts[2].y ? pts[2].y : pts[0].y;
  return 1;
}

/* ---- the interpreter ----------------------------------------------------------------------------------------- */
static void set_limits(ctx_t* c) {
  g_limits_on = 1; /* equivalent to cairo's "only when the stroke's extents leave the surface": see the header */
  g_l1.x = 0; g_l1.y = 0; g_l2.x = c->w * FIXED_ONE; g_l2.y = c->h * FIXED_ONE;
}

static int color_byte(double g) { /* _cairo_color_double_to_short, then the top byte */
  if (g < 0) g = 0;
  if (g > 1) g = 1;
  unsigned short s = (unsigned short)(g * 65535.0 + 0.5);
  return s >> 8;
}

static int run(const double* s, int n, int w, int h, unsigned char* out, int keep) {
  ctx_t c;
  memset(&c, 0, sizeof(c));
  c.w = w; c.h = h; c.fb = out;
  if (!keep) memset(out, 0, (size_t)w * h);
  mat_init(&c.gs.ctm, 1, 0, 0, 1, 0, 0);
  mat_init(&c.gs.inv, 1, 0, 0, 1, 0, 0);
  c.gs.lw = 2.0;
  c.gs.grey = 0;
  path_reset(&c);
  int i = 0, rc = 0;
  mat_t t;
  while (i < n && rc == 0) {
    int op = (int)s[i++];
    if (op == CM_END) break;
    switch (op) {
      case CM_SAVE: if (c.sp < 16) c.stack[c.sp++] = c.gs; else rc = -3; break;
      case CM_RESTORE: if (c.sp > 0) c.gs = c.stack[--c.sp]; else rc = -3; break;
      case CM_SCALE:
        mat_init(&t, s[i], 0, 0, s[i + 1], 0, 0); mat_mul(&c.gs.ctm, &t, &c.gs.ctm);
        mat_init(&t, 1 / s[i], 0, 0, 1 / s[i + 1], 0, 0); mat_mul(&c.gs.inv, &c.gs.inv, &t);
        i += 2; break;
      case CM_TRANSLATE:
        mat_init(&t, 1, 0, 0, 1, s[i], s[i + 1]); mat_mul(&c.gs.ctm, &t, &c.gs.ctm);
        mat_init(&t, 1, 0, 0, 1, -s[i], -s[i + 1]); mat_mul(&c.gs.inv, &c.gs.inv, &t);
        i += 2; break;
      case CM_ROTATE: {
        double sn = sin(s[i]), cs = cos(s[i]);
        mat_init(&t, cs, sn, -sn, cs, 0, 0); mat_mul(&c.gs.ctm, &t, &c.gs.ctm);
        sn = sin(-s[i]); cs = cos(-s[i]);
        mat_init(&t, cs, sn, -sn, cs, 0, 0); mat_mul(&c.gs.inv, &c.gs.inv, &t);
        i += 1; break;
      }
      case CM_LINE_WIDTH: c.gs.lw = s[i]; i += 1; break;
      case CM_GREY: c.gs.grey = color_byte(s[i]); i += 1; break;
      case CM_MOVE_TO: path_move_to_fixed(&c, user_to_fixed(&c, s[i], s[i + 1])); i += 2; break;
      case CM_LINE_TO: path_line_to_fixed(&c, user_to_fixed(&c, s[i], s[i + 1])); i += 2; break;
      case CM_CLOSE: path_close(&c); break;
      case CM_ARC: {
        double xc = s[i], yc = s[i + 1], r = s[i + 2], a1 = s[i + 3], a2 = s[i + 4];
        i += 5;
        if (r <= 0.0) { path_line_to_fixed(&c, user_to_fixed(&c, xc, yc)); path_line_to_fixed(&c, user_to_fixed(&c, xc, yc)); break; }
        while (a2 < a1) a2 += 2 * M_PI;
        path_line_to_fixed(&c, user_to_fixed(&c, xc + r * cos(a1), yc + r * sin(a1)));
        arc_in_direction(&c, xc, yc, r, a1, a2);
        break;
      }
      case CM_RECT: {
        double x = s[i], y = s[i + 1], ww = s[i + 2], hh = s[i + 3];
        i += 4;
        path_move_to_fixed(&c, user_to_fixed(&c, x, y));
        /* cairo_rel_line_to: the DISTANCE goes through the matrix and is added to the fixed current point */
        double dx, dy;
        pt_t p;
        dx = ww; dy = 0; mat_distance(&c.gs.ctm, &dx, &dy);
        p.x = c.cur.x + fx_from_double(dx); p.y = c.cur.y + fx_from_double(dy); path_line_to_fixed(&c, p);
        dx = 0; dy = hh; mat_distance(&c.gs.ctm, &dx, &dy);
        p.x = c.cur.x + fx_from_double(dx); p.y = c.cur.y + fx_from_double(dy); path_line_to_fixed(&c, p);
        dx = -ww; dy = 0; mat_distance(&c.gs.ctm, &dx, &dy);
        p.x = c.cur.x + fx_from_double(dx); p.y = c.cur.y + fx_from_double(dy); path_line_to_fixed(&c, p);
        path_close(&c);
        break;
      }
      case CM_CURVE_TO: {
        pt_t b = user_to_fixed(&c, s[i], s[i + 1]), cc = user_to_fixed(&c, s[i + 2], s[i + 3]),
             d = user_to_fixed(&c, s[i + 4], s[i + 5]);
        i += 6;
        path_curve_to_fixed(&c, b, cc, d);
        break;
      }
      case CM_NEW_PATH: path_reset(&c); break;
      case CM_STROKE: {
        poly_reset();
        if (c.gs.lw > 0.0 && c.nops) {
          box_t bx[16];
          int nb = c.rectilinear ? stroke_rectilinear_boxes(&c, bx, 16) : -1;
          if (nb >= 0) {
            boxes_render(&c, bx, nb, c.gs.grey);
          } else {
            set_limits(&c);
            stroke_to_polygon(&c);
            tor_render(&c, c.gs.grey);
          }
        }
        path_reset(&c);
        break;
      }
      case CM_FILL: {
        poly_reset();
        box_t b;
        if (c.nops && path_is_box(&c, &b)) {
          boxes_render(&c, &b, 1, c.gs.grey);
        } else if (c.nops) {
          set_limits(&c);
          fill_to_polygon(&c);
          tor_render(&c, c.gs.grey);
        }
        path_reset(&c);
        break;
      }
      case CM_PAINT: memset(out, c.gs.grey, (size_t)w * h); break;
      default: rc = -1; break;
    }
  }
  free(c.ops);
  return rc;
}

int cm_run(const double* script, int n, int w, int h, unsigned char* out) { return run(script, n, w, h, out, 0); }
int cm_run_over(const double* script, int n, int w, int h, unsigned char* out) { return run(script, n, w, h, out, 1); }

int cm_last_polygon(int32_t* out, int cap) {
  for (int i = 0; i < g_nedges && i < cap; i++) {
    const edge_t* e = &g_edges[i];
    int32_t* o = out + 7 * i;
    o[0] = e->p1.x; o[1] = e->p1.y; o[2] = e->p2.x; o[3] = e->p2.y; o[4] = e->top; o[5] = e->bottom; o[6] = e->dir;
  }
  return g_nedges;
}
