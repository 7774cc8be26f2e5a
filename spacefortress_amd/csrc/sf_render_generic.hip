// sf_render_generic.hip -- the image observation in ANY geometry the reference's constructor takes:
//   SSF_Env(scale, viewport, ls)  ENV:50-60  ->  sf.Game(width = int(vw * scale), height = int(vh * scale), viewport, lw = ls)
//   drawGameStateScaled           SRC/draw.cpp:256-270: scale(w / vw, h / vh), translate(-vx, -vy), line width ls user units
// sf_render.hip is the frame kernel of the DEFAULT geometry (scale .2, viewport (130, 80, 450, 460), ls 3: the trainer's,
// BASELINE configs[4]): a wave per env, the 90x92 surface, its tap periods, picture caches and tables all compile-time
// facts.  This file is the general renderer behind sf_set_image_geometry: one WORKGROUP per env, the W x H surface in
// dynamic LDS, every cairo_stroke of the reference's draw order (SRC/draw.cpp:227-254,266-268) rasterised in place, one object
// after the other, the way cairo's image backend does it (sf_tor.h / sf_tor_dev.h: the same code as the default geometry's
// kernel, the geometry a run-time argument), no caches, no shortcuts; then cv2.resize(.., (84, 84), INTER_AREA) with per-batch
// tap tables for W, H < 3 * 84 (OpenCV's resizeArea_ arithmetic, sf_image.cpp).  It reads the state, not the draw records.
// Phases: all four waves copy the background in (16-byte pieces); wave 0 alone rasterises the objects, wave-synchronously, and
// the score's glyphs with a lane per pixel of their box; all four waves resample, four pixels and one 32-bit store per thread.
// Correctness in every geometry (tests/test_gpu_image.py: against frames the reference's own renderer drew in four of them);
// speed in the one the benchmark names.
#include <hip/hip_runtime.h>

#include "sf_internal.h"
#include "sf_raster.h"
#include "sf_tor_dev.h"

namespace {

constexpr int kThreads = 256;

struct d2_t {
  double x, y;
};
struct i4_t {
  int x, y, z, w;
};
#define G_CHUNK(group, s) (tile + sfl::chunk_offset(SF_G_##group, (s)))
#define G_LD(T, base, off) (*reinterpret_cast<const T*>((base) + (off)))

struct Seg4 {
  double ax, ay, bx, by;
};
// wireframe segments, SRC/wireframe.cpp:11-67: kind 0 ship, 1 fortress, 2 missile, 3 shell
__device__ __forceinline__ Seg4 wire_seg(int kind, int k) {
  if (kind == 0) return Seg4{k == 2 ? 0.0 : -18.0, k == 1 ? 18.0 : 0.0, k == 0 ? 18.0 : (k == 1 ? 0.0 : -18.0), k == 2 ? -18.0 : 0.0};
  if (kind == 1) return Seg4{k >= 2 ? 18.0 : 0.0, k == 0 ? 0.0 : (k == 3 ? 18.0 : -18.0), k == 0 ? 36.0 : (k == 3 ? 0.0 : 18.0), k == 0 ? 0.0 : (k == 1 ? -18.0 : 18.0)};
  if (kind == 2) return Seg4{0.0, 0.0, k == 0 ? -25.0 : -5.0, k == 0 ? 0.0 : (k == 1 ? 5.0 : -5.0)};
  return Seg4{k == 0 ? -8.0 : (k == 2 ? 16.0 : 0.0), k == 1 ? -6.0 : (k == 3 ? 6.0 : 0.0), k == 1 ? 16.0 : (k == 3 ? -8.0 : 0.0), k == 0 ? -6.0 : (k == 2 ? 6.0 : 0.0)};
}

struct Ctx {
  uint8_t* fb;
  uint32_t* tor;
  int W, H, tid;
  double sx, sy, vx, vy, lw;  // scale_x = W / vp_w, scale_y = H / vp_h (SRC/draw.cpp:70-71), viewport origin, line width
  const double* trig;         // cos, sin of deg2rad(k)
  const double* arcs;         // sf_arc_table
  __device__ __forceinline__ sftd::Ctx tc() const { return sftd::Ctx{tor, fb, W, H, tid, sftd::kMaxQuadsBig}; }
  __device__ __forceinline__ static void order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // drawWireFrame (SRC/draw.cpp:82-100): one cairo_stroke of the wireframe's lines
  __device__ void wireframe(int kind, int deg, double px, double py) const {
    deg = deg < 0 ? 0 : (deg > 359 ? 359 : deg);
    const int n = (kind == 0 || kind == 2) ? 3 : 4;
    const sft::Affine v = sft::view_matrix(sx, sy, vx, vy);
    const double2 cs = *reinterpret_cast<const double2*>(trig + 2 * deg);
    const sft::Affine m = sft::object_matrix(v, px, py, cs.x, cs.y);
    if (kind == 1 && m.xy == 0.0 && m.yx == 0.0) {  // the rectilinear stroker (heading 0): boxes, exact area
      fort_boxes(m);
      return;
    }
    sft::Quad q = {};
    if (tid < n) {
      const Seg4 g = wire_seg(kind, tid);
      int x1, y1, x2, y2;
      sft::to_device(m, g.ax, g.ay, &x1, &y1);
      sft::to_device(m, g.bx, g.by, &x2, &y2);
      q = sft::stroke_quad(x1, y1, x2, y2, sx, sy, lw / 2);
    }
    sftd::raster(tc(), q, tid < n, 0, kind == 1 ? sftd::kKindFort : (kind == 3 ? sftd::kKindShell : sftd::kKindLines3), 255);
  }
  // the fortress at heading 0 under a matrix without rotation: cairo strokes the four axis-aligned lines as BOXES
  // (cairo-path-stroke-boxes.c) and fills their union through the box converter: exact area, alpha = c - (c >> 8)
  __device__ void fort_boxes(const sft::Affine& m) const {
    const int hx = sft::fx_from_double(fabs(m.xx) * lw / 2.0), hy = sft::fx_from_double(fabs(m.yy) * lw / 2.0);
    int bx[4][4];
    int X0 = 1 << 30, Y0 = 1 << 30, X1 = -(1 << 30), Y1 = -(1 << 30);
    for (int k = 0; k < 4; k++) {
      const Seg4 g = wire_seg(1, k);
      int x1, y1, x2, y2;
      sft::to_device(m, g.ax, g.ay, &x1, &y1);
      sft::to_device(m, g.bx, g.by, &x2, &y2);
      if (y1 == y2) { y1 -= hy; y2 += hy; } else { x1 -= hx; x2 += hx; }
      bx[k][0] = min(x1, x2); bx[k][1] = min(y1, y2); bx[k][2] = max(x1, x2); bx[k][3] = max(y1, y2);
      X0 = min(X0, bx[k][0]); Y0 = min(Y0, bx[k][1]); X1 = max(X1, bx[k][2]); Y1 = max(Y1, bx[k][3]);
    }
    const int px0 = max(X0 >> 8, 0), py0 = max(Y0 >> 8, 0), px1 = min((X1 + 255) >> 8, W), py1 = min((Y1 + 255) >> 8, H);
    const int bw = px1 - px0, n = bw > 0 && py1 > py0 ? bw * (py1 - py0) : 0;
    for (int i = tid; i < n; i += 64) {
      const int ry = i / bw, px = px0 + (i - ry * bw), py = py0 + ry;
      // |A u B u C u D| = sum of the four - the overlaps of the pairs that can overlap: the bar through the upright
      // (0, 2) and the two corners (1, 2), (2, 3)
      auto ov = [&](int a0, int a1, int b0, int b1, int c0, int c1) { const int o = min(min(a1, b1), c1) - max(max(a0, b0), c0); return o > 0 ? o : 0; };
      const int PX0 = px << 8, PX1 = PX0 + 256, PY0 = py << 8, PY1 = PY0 + 256;
      long long area = 0;
      for (int k = 0; k < 4; k++)
        area += (long long)ov(bx[k][0], bx[k][2], PX0, PX1, PX0, PX1) * ov(bx[k][1], bx[k][3], PY0, PY1, PY0, PY1);
      const int pr[3][2] = {{0, 2}, {1, 2}, {2, 3}};
      for (int j = 0; j < 3; j++) {
        const int a = pr[j][0], b = pr[j][1];
        area -= (long long)ov(bx[a][0], bx[a][2], bx[b][0], bx[b][2], PX0, PX1) * ov(bx[a][1], bx[a][3], bx[b][1], bx[b][3], PY0, PY1);
      }
      const int al = sft::box_area_to_alpha(area);
      if (al) {
        uint8_t* p = fb + py * W + px;
        *p = (uint8_t)sft::lerp8(255, al, *p);
      }
    }
    order();
  }
  // drawExplosion (SRC/draw.cpp:116-145): 7 rings of twelve 10-degree arcs, each its own stroke, then the radius-7 circle.
  // Every curve is flattened as cairo flattens it at THIS geometry's scale (sft::flatten_faces: one piece per arc at scale .2,
  // two at .4; eight per half circle, or sixteen): four arcs of up to four pieces per call, then the circle's up to 32 pieces.
  __device__ void explosion(double cx, double cy) const {
    const sft::Affine v = sft::view_matrix(sx, sy, vx, vy);
    const double hw = (double)(float)lw / 2;
    for (int ring = 0; ring < 7; ring++)
      for (int chunk = 0; chunk < 3; chunk++) {
        sft::Quad q = {};
        bool valid = false;
        if (tid < 16) {
          const double* kp = arcs + 8 * (12 * ring + 4 * chunk + (tid >> 2));
          int px[6], py[6], tx[6], ty[6];
          const int n = sft::flatten_faces(sft::arc_knots(v, cx, cy, sft::ArcK{kp[0], kp[1], kp[2], kp[3], kp[4], kp[5], kp[6], kp[7]}), px, py, tx, ty, 5);
          const int p = tid & 3;
          if (p < n - 1) {
            valid = true;
            q = sft::faces_quad(px[p], py[p], tx[p], ty[p], px[p + 1], py[p + 1], tx[p + 1], ty[p + 1], sx, sy, hw);
          }
        }
        sftd::raster(tc(), q, valid, tid & ~3, sftd::kKindSingle, 15 + 8 * ring < 60 ? 191 : 128);
      }
    sft::Quad q = {};
    bool valid = false;
    int m0 = 0;
    {
      int px[18], py[18], tx[18], ty[18];
      const double* kp = arcs + 8 * (84 + ((tid >> 4) & 1));
      const int n = sft::flatten_faces(sft::arc_knots(v, cx, cy, sft::ArcK{kp[0], kp[1], kp[2], kp[3], kp[4], kp[5], kp[6], kp[7]}), px, py, tx, ty, 17);
      const int p = tid & 15;
      if (tid < 32 && p < n - 1) {
        valid = true;
        q = sft::faces_quad(px[p], py[p], tx[p], ty[p], px[p + 1], py[p + 1], tx[p + 1], ty[p + 1], sx, sy, hw);
      }
      m0 = __builtin_amdgcn_readfirstlane(n - 1);  // (lane 0 flattened the first half)
    }
    sftd::raster<8>(tc(), q, valid, 0, sftd::kKindRing | (m0 << 8), 191);
  }
  // the score (drawScore, SRC/draw.cpp:161-173): "%07d", grey .5 through the coverage of this geometry's glyph atlas
  // (sf_glyphs.h, sf_set_score_glyphs) -- a lane per pixel of the text's box --, or, without one, the seven-segment FALLBACK
  // (sf_raster.h: equal to no reference pixels) -- a lane per pixel composites the segments that touch it in the strokes' order
  __device__ void score(int pnts, const SfGlyphAtlas* G) const {
    if (G && G->gw) {  // (uniform)
      const uint32_t chars = sfg::score_chars(pnts);
      const int bx0 = max((int)G->x_min, 0), bx1 = min((int)G->x_max + 6 * G->advance + G->gw, W);
      const int by0 = max(G->y0, 0), by1 = min(G->y0 + G->gh, H);
      const int bw = bx1 - bx0, n = bw > 0 && by1 > by0 ? bw * (by1 - by0) : 0;
      for (int i = tid; i < n; i += 64) {
        const int ry = i / bw, px = bx0 + (i - ry * bw), py = by0 + ry;
        uint8_t* p = fb + py * W + px;
        *p = (uint8_t)sfg::text_pixel(G, chars, px, py, *p);
      }
      order();
      return;
    }
    // (the fallback: float64 here -- the segment model is ours, oracle/render_np.py evaluates it in float64, and this kernel has the time)
    const unsigned long long masks = sfr::score_masks(pnts);
    auto dx = [&](double x) { return (x - vx) * sx; };
    auto dy = [&](double y) { return (y - vy) * sy; };
    const double Wg = SF_TXT_W, Hg = SF_TXT_H, T = SF_TXT_T, m0 = 0.5 * (SF_TXT_H - SF_TXT_T), m1 = 0.5 * (SF_TXT_H + SF_TXT_T);
    const double sx0[7] = {0, Wg - T, Wg - T, 0, 0, 0, 0}, sx1[7] = {Wg, Wg, Wg, Wg, T, T, Wg};
    const double sy0[7] = {0, T, m1, Hg - T, m1, T, m0}, sy1[7] = {T, m0, Hg - T, Hg, Hg - T, m0, m1};
    const double tx0 = dx((double)SF_TXT_X0 + SF_TXT_PAD), tx1 = dx((double)SF_TXT_X0 + 6.0 * SF_TXT_ADV + SF_TXT_PAD + SF_TXT_W);
    const double ty0 = dy((double)SF_TXT_TOP), ty1 = dy((double)SF_TXT_TOP + SF_TXT_H);
    const int bx0 = max((int)floor(tx0), 0), by0 = max((int)floor(ty0), 0), bx1 = min((int)ceil(tx1), W), by1 = min((int)ceil(ty1), H);
    const int bw = bx1 - bx0, n = bw > 0 && by1 > by0 ? bw * (by1 - by0) : 0;
    for (int i = tid; i < n; i += 64) {
      const int ry = i / bw, px = bx0 + (i - ry * bw), py = by0 + ry;
      const double fpx = (double)px, fpy = (double)py;
      uint8_t* p = fb + py * W + px;
      int d = *p;
      for (int cell = 0; cell < 7; cell++) {
        const double gx = (double)SF_TXT_X0 + (double)SF_TXT_ADV * (double)cell + SF_TXT_PAD, gy = SF_TXT_TOP;
        const unsigned bits = (unsigned)(masks >> (7 * cell)) & 0x7Fu;
#pragma unroll
        for (int seg = 0; seg < 7; seg++) {
          if (!((bits >> seg) & 1u)) continue;
          // the model's own arithmetic: the segment's rectangle clipped to the pixel, its area (render_np.pixel_area)
          const double ox = fmin(dx(gx + sx1[seg]), fpx + 1.0) - fmax(dx(gx + sx0[seg]), fpx);
          const double oy = fmin(dy(gy + sy1[seg]), fpy + 1.0) - fmax(dy(gy + sy0[seg]), fpy);
          if (ox > 0.0 && oy > 0.0) {
            const int m = (int)(fmin(ox * oy, 1.0) * 255.0 + 0.5);
            if (m > 0) d = sfr::over_un8(d, 128, m);
          }
        }
      }
      *p = (uint8_t)d;
    }
    order();
  }
  // drawVlner (SRC/draw.cpp:207-225): cairo_rectangle(x, y, w, h) + cairo_fill: the box converter
  __device__ void rect(double x, double y, double w, double h, int grey) const {
    const sft::Affine v = sft::view_matrix(sx, sy, vx, vy);
    int x0, y0;
    sft::to_device(v, x, y, &x0, &y0);
    int x1 = x0 + sft::fx_from_double(v.xx * w + v.xy * 0.0), y1 = y0 + sft::fx_from_double(v.yx * 0.0 + v.yy * h);
    if (x1 < x0) { const int t = x0; x0 = x1; x1 = t; }
    if (y1 < y0) { const int t = y0; y0 = y1; y1 = t; }
    const int px0 = max(x0 >> 8, 0), py0 = max(y0 >> 8, 0), px1 = min((x1 + 255) >> 8, W), py1 = min((y1 + 255) >> 8, H);
    const int bw = px1 - px0, n = bw > 0 && py1 > py0 ? bw * (py1 - py0) : 0;
    for (int i = tid; i < n; i += 64) {
      const int ry = i / bw, px = px0 + (i - ry * bw), py = py0 + ry;
      const int ox = min(x1, (px + 1) << 8) - max(x0, px << 8), oy = min(y1, (py + 1) << 8) - max(y0, py << 8);
      if (ox > 0 && oy > 0) {
        const int al = sft::box_area_to_alpha((long long)ox * oy);
        if (al) {
          uint8_t* p = fb + py * W + px;
          *p = (uint8_t)sft::lerp8(grey, al, *p);
        }
      }
    }
    order();
  }
};

}  // namespace

struct SfGenericArgs {
  const unsigned char* state;
  int n_envs, W, H;
  double sx, sy, vx, vy, lw;
  const double* trig;
  const double* arcs;
  const uint8_t* bg;      // W * H bytes: the hexagons on black (sf_image.cpp: sf_image_background_geom)
  const uint32_t* tabs;   // resize != 0: 8 words per destination column, then per row: first, count, 4 weights, 2 pad
  uint8_t* out;
  size_t out_stride;
  int resize;
  const SfGlyphAtlas* glyphs;  // the score text's glyph atlas for THIS geometry (sf_glyphs.h); null or gw == 0: the fallback
};

__global__ __launch_bounds__(kThreads) void sf_render_generic_kernel(SfGenericArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t g_fb[];
  __shared__ double mtab[SF_NSLOT][3];
  __shared__ __attribute__((aligned(16))) uint32_t torw[sftd::kLdsWordsBig];
  const int tid = threadIdx.x, env = blockIdx.x;
  const int W = a.W, H = a.H;
  const Ctx C{g_fb, torw, W, H, tid, a.sx, a.sy, a.vx, a.vy, a.lw, a.trig, a.arcs};
  for (int i = tid; i < sftd::kLdsWordsBig; i += kThreads) torw[i] = 0u;
  const unsigned char* tile = a.state + (long)(env >> 6) * sfl::kTileBytes;
  const int l = env & 63, o16 = l * 16;
  const d2_t sp = G_LD(d2_t, G_CHUNK(ship_pos, 0), o16);
  const i4_t tb = G_LD(i4_t, G_CHUNK(timers_b, 0), o16);
  const i4_t sc = G_LD(i4_t, G_CHUNK(score, 0), o16);
  const i4_t mi = G_LD(i4_t, G_CHUNK(misc, 0), o16);
  const i4_t sm = G_LD(i4_t, G_CHUNK(small, 0), o16);
  const int ship_angle = (int)(int16_t)(sm.x & 0xFFFF), fort_angle = (int)(int16_t)((unsigned)sm.x >> 16);
  const unsigned flags = ((unsigned)sm.y >> 16) & 0xFFu;
  const unsigned mmask = (unsigned)mi.z & SF_MASK_LOW, smask = (unsigned)mi.w & SF_MASK_LOW, n_pool = (unsigned)mi.z >> SF_MPOOL_SHIFT;
  const int pnts = (int)__int_as_float(sc.x), vlner = sc.z & 0xFFF;
  // the background: both hexagons stroked on black (SRC/draw.cpp:230-231,262-263)
  // (16 bytes per thread and piece, four pieces in flight: left byte by byte, every iteration waits out its own L2 round
  //  trip -- 64 of them, a hundred microseconds per frame.  The host pads the background to whole pieces.)
  {
    const uint4* src = reinterpret_cast<const uint4*>(a.bg);
    uint4* dst = reinterpret_cast<uint4*>(g_fb);
    const int n16 = (W * H + 15) >> 4;
    for (int base = 0; base < n16; base += 4 * kThreads) {
      uint4 v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = src[min(base + k * kThreads + tid, n16 - 1)];
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (base + k * kThreads + tid < n16) dst[base + k * kThreads + tid] = v[k];
    }
  }
  // this env's missiles, out of the tile's pool, filed by slot (the reference draws in slot order, SRC/draw.cpp:243-247)
  for (unsigned e = tid; e < n_pool; e += kThreads) {
    const unsigned meta = G_LD(uint32_t, G_CHUNK(missile_meta, 0), e * 4u);
    if (SF_MM_OWNER(meta) == (unsigned)l) {
      const d2_t m = G_LD(d2_t, G_CHUNK(missile_pos, 0), e * 16u);
      double* t = mtab[SF_MM_SLOT(meta)];
      t[0] = m.x;
      t[1] = m.y;
      t[2] = (double)SF_MM_ANGLE(meta);
    }
  }
  __syncthreads();
  if (tid < 64) {  // ---- wave 0 composites: ship (:233-237), fortress (:238-242)
  if (flags & SF_FL_SHIP_ALIVE) C.wireframe(0, ship_angle, sp.x, sp.y);
  else C.explosion(sp.x, sp.y);
  if (flags & SF_FL_FORT_ALIVE) C.wireframe(1, fort_angle, sfc::fort_x, sfc::fort_y);
  else C.explosion(sfc::fort_x, sfc::fort_y);
  // missiles (:243-247), shells (:248-253: only once clear of the fortress; drawWireFrame takes the heading as an int)
  for (int s = 0; s < SF_NSLOT; s++)
    if ((mmask >> s) & 1u) C.wireframe(2, (int)mtab[s][2], mtab[s][0], mtab[s][1]);
  for (int s = 0; s < SF_NSLOT; s++)
    if ((smask >> s) & 1u) {
      const d2_t p = G_LD(d2_t, G_CHUNK(shell_pos, s), o16), v = G_LD(d2_t, G_CHUNK(shell_vel, s), o16);
      const double ddx = p.x - sfc::fort_x, ddy = p.y - sfc::fort_y;
      if (sqrt(ddx * ddx + ddy * ddy) > 21.0) {
        double ang = atan2(v.y, v.x) * 180.0 / M_PI;
        if (ang < 0) ang += 360.0;
        C.wireframe(3, (int)ang, p.x, p.y);
      }
    }
  C.score(pnts, a.glyphs);
  // vulnerability bar (drawVlner, :205-225,268)
  {
    const bool kill = vlner > 10 && tb.w < sfc::vuln_time;
    C.rect(355.0 - 100, 335.0 + 187, 200.0, 10.0, 84);
    if (vlner > 0) C.rect(355.0 - 100, 335.0 + 187, (double)(20 * (vlner > 10 ? 10 : vlner)), 10.0, kill ? 255 : 168);
  }
  }  // (wave 0)
  __syncthreads();
  uint8_t* const frame_out = a.out + (size_t)env * a.out_stride;
  if (!a.resize) {
    // the raw frame leaves in 4-byte words at 4-byte aligned addresses (a frame of W * H bytes need not start on one: a few
    // bytes first): byte stores are one 64-byte write per wave instruction and were half of the launch
    const int head = (int)((0u - (unsigned)(uintptr_t)frame_out) & 3u), nw = (W * H - head) >> 2, tail = head + 4 * nw;
    if (tid < head) frame_out[tid] = g_fb[tid];
    for (int j = tid; j < nw; j += kThreads) {
      uint32_t v;
      __builtin_memcpy(&v, g_fb + head + 4 * j, 4);
      *reinterpret_cast<uint32_t*>(frame_out + head + 4 * j) = v;
    }
    if (tail + tid < W * H) frame_out[tail + tid] = g_fb[tail + tid];
    return;
  }
  // cv2.resize(frame, (84, 84), INTER_AREA): per source row buf = sum alpha * S in table order, sum (+)= beta * buf in table
  // order, saturate_cast<uchar> (round half to even) -- resizeArea_<uchar, float>.  Four destination pixels of a row per
  // thread, one aligned 32-bit store (the 84x84 frames are 16-byte aligned); the taps are loaded before the sums start.
  const uint4* ct4 = reinterpret_cast<const uint4*>(a.tabs);
  const uint4* rt4 = reinterpret_cast<const uint4*>(a.tabs + 8 * SF_OUT);
  for (int i = tid; i < SF_OUT * (SF_OUT / 4); i += kThreads) {
    const int oy = i / (SF_OUT / 4), ox0 = 4 * (i - oy * (SF_OUT / 4));
    const uint4 ra = rt4[2 * oy], rb = rt4[2 * oy + 1];  // first, count, b0, b1 | b2, b3, -, -
    uint4 ca[4], cb[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ca[p] = ct4[2 * (ox0 + p)];
      cb[p] = ct4[2 * (ox0 + p) + 1];
    }
    const int rf = (int)ra.x, rc = (int)ra.y;
    const float beta[4] = {__uint_as_float(ra.z), __uint_as_float(ra.w), __uint_as_float(rb.x), __uint_as_float(rb.y)};
    unsigned word = 0u;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int cf = (int)ca[p].x, cc = (int)ca[p].y;
      const float al[4] = {__uint_as_float(ca[p].z), __uint_as_float(ca[p].w), __uint_as_float(cb[p].x), __uint_as_float(cb[p].y)};
      float sum = 0.f;
      for (int k = 0; k < rc; k++) {
        const uint8_t* S = g_fb + (rf + k) * W + cf;
        float bsum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (j < cc) bsum += (float)S[j] * al[j];
        const float bk = k == 0 ? beta[0] : (k == 1 ? beta[1] : (k == 2 ? beta[2] : beta[3]));
        sum = k == 0 ? bk * bsum : sum + bk * bsum;
      }
      int v = (int)rintf(sum);
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      word |= (unsigned)v << (8 * p);
    }
    *reinterpret_cast<uint32_t*>(frame_out + oy * SF_OUT + ox0) = word;
  }
}

hipError_t sf_launch_render_generic(const unsigned char* state, int n_envs, int W, int H, double sx, double sy, double vp_x, double vp_y,
                                    double line_w, const double* trig, const double* arcs, const uint8_t* bg, const uint32_t* tabs,
                                    uint8_t* out, size_t out_stride, int resize, const SfGlyphAtlas* glyphs, hipStream_t stream) {
  if (n_envs <= 0) return hipSuccess;
  SfGenericArgs a{state, n_envs, W, H, sx, sy, vp_x, vp_y, line_w, trig, arcs, bg, tabs, out, out_stride, resize, glyphs};
  const size_t lds = ((size_t)W * H + 15) & ~(size_t)15;
  hipLaunchKernelGGL(sf_render_generic_kernel, dim3((unsigned)n_envs), dim3(kThreads), lds, stream, a);
  return hipGetLastError();
}
