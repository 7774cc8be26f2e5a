// sf_render_generic.hip -- the image observation in ANY geometry the reference's constructor takes:
//   SSF_Env(scale, viewport, ls)  ENV:50-60  ->  sf.Game(width = int(vw * scale), height = int(vh * scale), viewport, lw = ls)
//   drawGameStateScaled           SRC/draw.cpp:256-270: scale(s), translate(-vx, -vy), line width ls user units
// sf_render.hip is the frame kernel of the DEFAULT geometry (scale .2, viewport (130, 80, 450, 460), ls 3: the trainer's,
// BASELINE configs[4]): a wave per env, the 90x92 surface, its tap periods, picture caches and stroke tables all compile-time
// facts.  This file is the general renderer behind sf_set_image_geometry: one WORKGROUP per env, the W x H surface in
// dynamic LDS, every stroke of the reference's draw order (SRC/draw.cpp:227-254,266-268) composited in place one after the
// other with the same coverage model (sf_cover.h) and 8-bit OVER arithmetic (sf_raster.h), no caches, no shortcuts; then
// cv2.resize(.., (84, 84), INTER_AREA) with per-batch tap tables for W, H < 3 * 84 (OpenCV's resizeArea_ arithmetic,
// sf_image.cpp).  It reads the state, not the draw records.  Phases: all four waves copy the background in (16-byte pieces);
// wave 0 alone composites the strokes, one after the other, wave-synchronously (no workgroup barrier per stroke) and the
// score's glyphs with a lane per pixel of their box; all four waves resample, four pixels and one 32-bit store per thread.
// Measured (tools/geometry_probe.py, 16 384 envs, 125 x 130 surface): 0.96 ms per launch for the raw frame, 1.24 ms with the
// 84x84 image -- 14.8 k vector instructions per frame (tools/pmc_generic.sh) against 0.95 k in the default geometry's kernel,
// most of them the 85 strokes of a dead ship's explosion drawn afresh in every frame of its 30 (no cache here).  Correctness
// in every geometry; speed in the one the benchmark names.  tests/test_gpu_image.py compares it with oracle/render_np.py
// parametrised the same way.
#include <hip/hip_runtime.h>

#include "sf_cover.h"
#include "sf_internal.h"
#include "sf_raster.h"

namespace {

using namespace sfcov;
using sfr::cover_to_mask;

#include "sf_render_tables.h"  // kArcs[7][12], kGon[12], kSinCosDeg[360]

constexpr int kThreads = 256;

struct d2_t {
  double x, y;
};
struct i4_t {
  int x, y, z, w;
};
#define G_CHUNK(group, s) (tile + sfl::chunk_offset(SF_G_##group, (s)))
#define G_LD(T, base, off) (*reinterpret_cast<const T*>((base) + (off)))

struct Ctx {
  uint8_t* fb;
  int W, H, tid;
  float vx, vy, sc, half_lw;  // user -> device: (x - vx) * sc; half the line width in user units
  __device__ __forceinline__ float dx(float x) const { return (x - vx) * sc; }
  __device__ __forceinline__ float dy(float y) const { return (y - vy) * sc; }

  // one stroke, composited OVER the surface by wave 0 (a lane per pixel of its bounding box)
  __device__ void stroke(const Quad& q, int grey) const {
    const float fx0 = fminf(fminf(q.x[0], q.x[1]), fminf(q.x[2], q.x[3])), fx1 = fmaxf(fmaxf(q.x[0], q.x[1]), fmaxf(q.x[2], q.x[3]));
    const float fy0 = fminf(fminf(q.y[0], q.y[1]), fminf(q.y[2], q.y[3])), fy1 = fmaxf(fmaxf(q.y[0], q.y[1]), fmaxf(q.y[2], q.y[3]));
    if (fx1 > 0.f && fy1 > 0.f && fx0 < (float)W && fy0 < (float)H) {  // (uniform: the quad is)
      const int x0 = (int)floorf(fmaxf(fx0, 0.f)), y0 = (int)floorf(fmaxf(fy0, 0.f));
      const int x1 = (int)ceilf(fminf(fx1, (float)W)), y1 = (int)ceilf(fminf(fy1, (float)H));
      const int bw = x1 - x0, n = bw * (y1 - y0);
      const Slopes sl = quad_slopes(q);
      for (int i = tid; i < n; i += 64) {
        const int ry = i / bw, px = x0 + (i - ry * bw), py = y0 + ry;
        const int m = cover_to_mask(quad_cover(q, sl, (float)px, (float)py));
        if (m > 0) {
          uint8_t* p = fb + py * W + px;
          *p = (uint8_t)sfr::over_un8(*p, grey, m);
        }
      }
    }
    order();
  }
  // the strokes are composited by ONE wave (tid = its lane): LDS is in order per wave, the fences pin the compiler
  __device__ __forceinline__ static void order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // a wireframe segment (ax, ay) - (bx, by) under translate(pos) rotate(angle): a rectangle, butt caps
  __device__ void line(float ax, float ay, float bx, float by, float ca, float sa, float px, float py, int grey) const {
    const float ux = bx - ax, uy = by - ay;
    const float inv = half_lw / sqrtf(ux * ux + uy * uy);
    const float nx = -uy * inv, ny = ux * inv;
    const float lx[4] = {ax + nx, bx + nx, bx - nx, ax - nx}, ly[4] = {ay + ny, by + ny, by - ny, ay - ny};
    Quad q;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      q.x[k] = dx(px + ca * lx[k] - sa * ly[k]);
      q.y[k] = dy(py + sa * lx[k] + ca * ly[k]);
    }
    stroke(q, grey);
  }
  // a filled axis-aligned rectangle in user units
  __device__ void rect(float x0, float y0, float x1, float y1, int grey) const {
    Quad q;
    q.x[0] = dx(x0); q.y[0] = dy(y0);
    q.x[1] = dx(x1); q.y[1] = dy(y0);
    q.x[2] = dx(x1); q.y[2] = dy(y1);
    q.x[3] = dx(x0); q.y[3] = dy(y1);
    stroke(q, grey);
  }
  __device__ void wireframe(const float (*lines)[4], int n, int deg, float px, float py) const {
    deg = deg < 0 ? 0 : (deg > 359 ? 359 : deg);
    const float sa = kSinCosDeg[deg][0], ca = kSinCosDeg[deg][1];
    for (int k = 0; k < n; k++) line(lines[k][0], lines[k][1], lines[k][2], lines[k][3], ca, sa, px, py, 255);
  }
  // drawExplosion (SRC/draw.cpp:145-175): 7 rings (radius 15 + 8 i) of twelve 10-degree arcs, each its own stroke -- one
  // chord quad between radius -/+ half the line width --, then one radius-7 circle: the ring between two regular 12-gons
  __device__ void explosion(float cx, float cy) const {
    for (int ring = 0; ring < 7; ring++) {
      const float radius = 15.f + 8.f * (float)ring, ri = radius - half_lw, ro = radius + half_lw;
      const int grey = radius < 60.f ? 191 : 128;
      for (int k = 0; k < 12; k++) {
        const ArcCS t = kArcs[ring][k];
        Quad q;
        q.x[0] = dx(cx + ri * t.c0); q.y[0] = dy(cy + ri * t.s0);
        q.x[1] = dx(cx + ro * t.c0); q.y[1] = dy(cy + ro * t.s0);
        q.x[2] = dx(cx + ro * t.c1); q.y[2] = dy(cy + ro * t.s1);
        q.x[3] = dx(cx + ri * t.c1); q.y[3] = dy(cy + ri * t.s1);
        stroke(q, grey);
      }
    }
    const float gx = dx(cx), gy = dy(cy), ro = (7.f + half_lw) * sc, ri = (7.f - half_lw) * sc;
    const int x0 = max((int)floorf(gx - ro), 0), y0 = max((int)floorf(gy - ro), 0);
    const int x1 = min((int)ceilf(gx + ro), W), y1 = min((int)ceilf(gy + ro), H);
    const int bw = x1 - x0, n = bw > 0 && y1 > y0 ? bw * (y1 - y0) : 0;
    for (int i = tid; i < n; i += 64) {
      const int ry = i / bw, px = x0 + (i - ry * bw), py = y0 + ry;
      const float area = gon(gx, gy, ro, (float)px, (float)py) - (ri > 0.f ? gon(gx, gy, ri, (float)px, (float)py) : 0.f);
      const int m = cover_to_mask(fmaxf(area, 0.f));
      if (m > 0) {
        uint8_t* p = fb + py * W + px;
        *p = (uint8_t)sfr::over_un8(*p, 191, m);
      }
    }
    order();
  }
  // the score (drawScore, SRC/draw.cpp:190-203): "%07d", grey .5, seven-segment glyphs (sf_raster.h) -- a lane per pixel of the
  // text's box composites the segments that touch it in the strokes' order (cell by cell, A..G): what one stroke after the
  // other gives, without 42 strokes.  A segment is an axis-aligned rectangle: its coverage of a pixel is the overlap's area.
  __device__ void score(int pnts) const {
    const unsigned long long masks = sfr::score_masks(pnts);
    const float Wg = SF_TXT_W, Hg = SF_TXT_H, T = SF_TXT_T, m0 = 0.5f * (SF_TXT_H - SF_TXT_T), m1 = 0.5f * (SF_TXT_H + SF_TXT_T);
    const float sx0[7] = {0, Wg - T, Wg - T, 0, 0, 0, 0}, sx1[7] = {Wg, Wg, Wg, Wg, T, T, Wg};
    const float sy0[7] = {0, T, m1, Hg - T, m1, T, m0}, sy1[7] = {T, m0, Hg - T, Hg, Hg - T, m0, m1};
    const float tx0 = dx(SF_TXT_X0 + SF_TXT_PAD), tx1 = dx(SF_TXT_X0 + 6.f * SF_TXT_ADV + SF_TXT_PAD + SF_TXT_W);
    const float ty0 = dy(SF_TXT_TOP), ty1 = dy(SF_TXT_TOP + SF_TXT_H);
    const int bx0 = max((int)floorf(tx0), 0), by0 = max((int)floorf(ty0), 0), bx1 = min((int)ceilf(tx1), W), by1 = min((int)ceilf(ty1), H);
    const int bw = bx1 - bx0, n = bw > 0 && by1 > by0 ? bw * (by1 - by0) : 0;
    for (int i = tid; i < n; i += 64) {
      const int ry = i / bw, px = bx0 + (i - ry * bw), py = by0 + ry;
      const float fpx = (float)px, fpy = (float)py;
      uint8_t* p = fb + py * W + px;
      int d = *p;
      for (int cell = 0; cell < 7; cell++) {
        const float gx = SF_TXT_X0 + SF_TXT_ADV * (float)cell + SF_TXT_PAD, gy = SF_TXT_TOP;
        if (dx(gx + Wg) <= fpx || dx(gx) >= fpx + 1.f) continue;
        const unsigned bits = (unsigned)(masks >> (7 * cell)) & 0x7Fu;
#pragma unroll
        for (int seg = 0; seg < 7; seg++) {
          if (!((bits >> seg) & 1u)) continue;
          const float ox = fminf(dx(gx + sx1[seg]), fpx + 1.f) - fmaxf(dx(gx + sx0[seg]), fpx);
          const float oy = fminf(dy(gy + sy1[seg]), fpy + 1.f) - fmaxf(dy(gy + sy0[seg]), fpy);
          if (ox > 0.f && oy > 0.f) {
            const int m = cover_to_mask(ox * oy);
            if (m > 0) d = sfr::over_un8(d, 128, m);
          }
        }
      }
      *p = (uint8_t)d;
    }
    order();
  }
  // area of the regular 12-gon (centre (gx, gy), circumradius r, device pixels) inside the pixel at (px, py)
  __device__ static float gon(float gx, float gy, float r, float px, float py) {
    float s = 0.f;
    float x0 = gx + r * kGon[0][0] - px, y0 = gy + r * kGon[0][1] - py;
#pragma unroll 1
    for (int k = 1; k <= 12; k++) {
      const float x1 = gx + r * kGon[k % 12][0] - px, y1 = gy + r * kGon[k % 12][1] - py;
      s += edge_term(x0, y0, x1, edge_slope(x0, y0, x1, y1));
      x0 = x1;
      y0 = y1;
    }
    return fabsf(s);
  }
};

__constant__ float kShipLines[3][4] = {{-18, 0, 18, 0}, {-18, 18, 0, 0}, {0, 0, -18, -18}};              // SRC/wireframe.cpp:11-67
__constant__ float kFortLines[4][4] = {{0, 0, 36, 0}, {0, -18, 18, -18}, {18, -18, 18, 18}, {18, 18, 0, 18}};
__constant__ float kMissileLines[3][4] = {{0, 0, -25, 0}, {0, 0, -5, 5}, {0, 0, -5, -5}};
__constant__ float kShellLines[4][4] = {{-8, 0, 0, -6}, {0, -6, 16, 0}, {16, 0, 0, 6}, {0, 6, -8, 0}};

}  // namespace

struct SfGenericArgs {
  const unsigned char* state;
  int n_envs, W, H;
  float vx, vy, sc, half_lw;
  const uint8_t* bg;      // W * H bytes: the hexagons on black (sf_image.cpp: sf_image_background_geom)
  const uint32_t* tabs;   // resize != 0: 8 words per destination column, then per row: first, count, 4 weights, 2 pad
  uint8_t* out;
  size_t out_stride;
  int resize;
};

__global__ __launch_bounds__(kThreads) void sf_render_generic_kernel(SfGenericArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t g_fb[];
  __shared__ float mtab[SF_NSLOT][3];
  const int tid = threadIdx.x, env = blockIdx.x;
  const int W = a.W, H = a.H;
  const Ctx C{g_fb, W, H, tid, a.vx, a.vy, a.sc, a.half_lw};
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
      float* t = mtab[SF_MM_SLOT(meta)];
      t[0] = (float)m.x;
      t[1] = (float)m.y;
      t[2] = (float)SF_MM_ANGLE(meta);
    }
  }
  __syncthreads();
  if (tid < 64) {  // ---- wave 0 composites: ship (:233-237), fortress (:238-242)
  if (flags & SF_FL_SHIP_ALIVE) C.wireframe(kShipLines, 3, ship_angle, (float)sp.x, (float)sp.y);
  else C.explosion((float)sp.x, (float)sp.y);
  if (flags & SF_FL_FORT_ALIVE) C.wireframe(kFortLines, 4, fort_angle, (float)sfc::fort_x, (float)sfc::fort_y);
  else C.explosion((float)sfc::fort_x, (float)sfc::fort_y);
  // missiles (:243-247), shells (:248-253: only once clear of the fortress; drawWireFrame takes the heading as an int)
  for (int s = 0; s < SF_NSLOT; s++)
    if ((mmask >> s) & 1u) C.wireframe(kMissileLines, 3, (int)mtab[s][2], mtab[s][0], mtab[s][1]);
  for (int s = 0; s < SF_NSLOT; s++)
    if ((smask >> s) & 1u) {
      const d2_t p = G_LD(d2_t, G_CHUNK(shell_pos, s), o16), v = G_LD(d2_t, G_CHUNK(shell_vel, s), o16);
      const double ddx = p.x - sfc::fort_x, ddy = p.y - sfc::fort_y;
      if (sqrt(ddx * ddx + ddy * ddy) > 21.0) {
        double ang = atan2(v.y, v.x) * 180.0 / M_PI;
        if (ang < 0) ang += 360.0;
        C.wireframe(kShellLines, 4, (int)ang, (float)p.x, (float)p.y);
      }
    }
  C.score(pnts);
  // vulnerability bar (drawVlner, :205-225,268)
  {
    const bool kill = vlner > 10 && tb.w < sfc::vuln_time;
    C.rect(255.f, 522.f, 455.f, 532.f, 84);
    if (vlner > 0) C.rect(255.f, 522.f, 255.f + 20.f * (float)(vlner > 10 ? 10 : vlner), 532.f, kill ? 255 : 168);
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

hipError_t sf_launch_render_generic(const unsigned char* state, int n_envs, int W, int H, double scale, double vp_x, double vp_y,
                                    double line_w, const uint8_t* bg, const uint32_t* tabs, uint8_t* out, size_t out_stride,
                                    int resize, hipStream_t stream) {
  if (n_envs <= 0) return hipSuccess;
  SfGenericArgs a{state, n_envs, W, H, (float)vp_x, (float)vp_y, (float)scale, (float)(line_w / 2), bg, tabs, out, out_stride, resize};
  const size_t lds = ((size_t)W * H + 15) & ~(size_t)15;
  hipLaunchKernelGGL(sf_render_generic_kernel, dim3((unsigned)n_envs), dim3(kThreads), lds, stream, a);
  return hipGetLastError();
}
