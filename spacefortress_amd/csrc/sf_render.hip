// sf_render.hip -- the image observation: SSF_Env._draw (ENV:203-206) + WrapPyTorch.observation
// (rl/envs.py:28-30) for a whole batch, on the device.
//
// The reference renders every env with cairo into a 90x92 RGB24 surface (grayscale=True, line width
// 3 user units = 0.6 px, SRC/draw.cpp:257-270), takes the grey channel (cv2.cvtColor, an identity
// on R=G=B) and the trainer's wrapper shrinks the frame to 84x84 with cv2.INTER_AREA.
//
// Here: ONE WAVEFRONT PER ENV, fed by the env's DRAW RECORD (sf_drawrec.h): what a frame decides from its env's state alone --
// which of the 148 backgrounds it starts from, whether the cached pictures apply, where the ship's box lies relative to the
// fortress's, the score's and the bar's -- was decided by the step kernel, where it costs a lane instead of a wave, and arrives
// as a 32-byte header through scalar loads; the objects' positions (float64) and headings arrive one per lane.  The kernel reads
// the state only for the shells (14 % of the frames have one).
// The 90x92 frame lives in LDS as bytes, starts as a copy of the background the record names (hexagons; score 0000000 / empty
// bar / the live fortress at its heading baked in; ten direct-to-LDS loads), and every cairo_stroke of the reference's draw order
// is rasterised the way cairo's image backend does it (sf_tor.h: 24.8 fixed-point corners, 15 sub-rows per pixel row or the whole
// row at once, the union of an object's lines, 8-bit lerp) by sf_tor_dev.h's lane arrangement: ship, missiles and shells in
// chunks of up to sixteen lines; an explosion ring by ring, twelve arcs at once, then its circle.  Whatever is a function of
// little is drawn once and copied afterwards: a dead ship's explosion (per env, keyed by where the ship died; with the score /
// bar box under it), and once per batch the live fortress at its 36 headings (an alpha map from the host, baked into
// backgrounds), the destroyed fortress's explosion, 1 024 scores and the bar's 12 states.  The 84x84 frame is built IN PLACE in
// the caller's buffer in HBM: it starts as the resampled background (host-made) and the wave re-evaluates INTER_AREA for exactly
// the output pixels that read a drawn object's box (out_box) -- a frame is a few dozen changed pixels on a static picture.  The
// frames whose ship just died (the expensive ones) are started first (pick_env).
//
// Pixel values: PINNED to the reference's own renderer (SRC/draw.cpp against cairo 1.16): tests/golden/frames holds frames it
// drew, oracle/cairo_model.c restates cairo's rasteriser bit for bit, and tests/test_gpu_image.py holds this kernel to both --
// all 92 rows: the score text is the reference's too, from a glyph atlas (sf_glyphs.h: FreeType's bitmaps as cairo blits them).
// Not pinned: cv2's INTER_AREA (absent from the image: OpenCV's published algorithm).
#include <hip/hip_runtime.h>

#include "sf_drawrec.h"
#include "sf_internal.h"
#include "sf_raster.h"
#include "sf_tor_dev.h"

// diagnostic builds only (tools/variant.py NAME -DSF_RENDER_SKIP=bits): bit 0 ship + fortress strokes, 1 missiles + shells,
// 2 score, 3 bar, 4 the resampling, 5 the coverage of the dense rounds, 6 their compositing, 8 the dead ship's explosion,
// 9 the fortress's explosion, 10 = explosion-cache misses take the hit path, 12 = no surface loads -- each bit removes that
// part (wrong pixels) so that its cost can be read off
#ifndef SF_RENDER_SKIP
#define SF_RENDER_SKIP 0
#endif
// diagnostic builds only: the frame kernel returns behind phase N (1 the prologue up to the barrier, 2 the cached explosion,
// 3 the fresh explosions, 41 .. 44 inside draw_strokes, 4 the strokes of ship / fortress / missiles, 5 the shells) --
// instruction counts of the phases by difference (tools/pmc_render_stops.sh)
#ifndef SF_RESAMPLE_QUADS
#define SF_RESAMPLE_QUADS 1
#endif
#ifndef SF_MERGE_SHELLS
#define SF_MERGE_SHELLS 1
#endif
#ifndef SF_RENDER_STOP
#define SF_RENDER_STOP 0
#endif
#ifndef SF_PREPASS_MAX_ENVS
#define SF_PREPASS_MAX_ENVS 6144 /* batches up to here get the explosion pre-pass (sf_explosion_kernel): measured break-even near 8 192 */
#endif

namespace {

constexpr int kFbBytes = SF_IMG_W * SF_IMG_H;          // 8280
constexpr int kFbWords = kFbBytes / 4;                 // 2070
constexpr int kFbPadWords = (SF_IMG_W * (SF_IMG_H + 1) + 3) / 4 + 1;  // one spare row for zero-weight taps
constexpr int kFbVec = kFbBytes / 16;                  // 517 (+ 8 bytes)
constexpr int kOutBytes = SF_OUT * SF_OUT;             // 7056 = 441 * 16

struct d2_t {
  double x, y;
};
struct i4_t {
  int x, y, z, w;
};

#define R_CHUNK(group, s) (tile + sfl::chunk_offset(SF_G_##group, (s)))
#define R_LD(T, base, off) (*reinterpret_cast<const T*>((base) + (off)))


// i / w and i % w for the pixel loops: i indexes a box of the 90x92 surface (i < 8 280), 1 <= w <= 92, `rw` = 1 / w
// (v_rcp_f32, hoisted out of the loop).  The general 32-bit division is two dozen instructions, once per pixel and loop:
// a sixth of what this kernel issued.  EXACT: (i + 0.5) / w lies at least 0.5 / w away from every integer and the float
// product is within (i / w) * 3 * 2^-24 of it, so truncation gives floor(i / w) whenever i * 6 * 2^-24 < 1, i.e. i < 2.7e6.
struct DivMod {
  int q, r;
};
__device__ __forceinline__ DivMod fast_divmod(int i, int w, float rw) {
  DivMod d;
  d.q = (int)(((float)i + 0.5f) * rw);
  d.r = i - d.q * w;
  return d;
}
__device__ __forceinline__ float recip_i(int w) { return __builtin_amdgcn_rcpf((float)w); }

using sfr::dev_x;
using sfr::dev_y;

__device__ __forceinline__ float bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

using sfd::Box;
using sfd::explosion_box;
using sfd::kReachX;
using sfd::kReachY;
using sfd::kHudTopRows;
using sfd::kHudBottomRows;
using sfd::kFpX0;
using sfd::kFpX1;
using sfd::kFpY0;
using sfd::kFpY1;

// pixel box of a fixed-point quad, clipped to the surface
__device__ __forceinline__ Box quad_box(const sft::Quad& q) {
  const int fx0 = min(min(q.x[0], q.x[1]), min(q.x[2], q.x[3])), fx1 = max(max(q.x[0], q.x[1]), max(q.x[2], q.x[3]));
  const int fy0 = min(min(q.y[0], q.y[1]), min(q.y[2], q.y[3])), fy1 = max(max(q.y[0], q.y[1]), max(q.y[2], q.y[3]));
  Box b;
  b.clear();
  if (fx1 > 0 && fy1 > 0 && fx0 < SF_IMG_W * 256 && fy0 < SF_IMG_H * 256) {
    b.x0 = max(fx0 >> 8, 0);
    b.y0 = max((sft::to_grid_y(fy0) - 1) / sft::kGridY, 0);  // (a sub-row's centre may lie a thirtieth of a pixel outside the corners)
    b.x1 = min((fx1 + 255) >> 8, SF_IMG_W);
    b.y1 = min((sft::to_grid_y(fy1) + sft::kGridY) / sft::kGridY, SF_IMG_H);
  }
  return b;
}

// Destination pixels of the 84x84 image that read source pixels of `b`, exactly: INTER_AREA gives destination
// column dx the source interval [dx * 15/14, (dx + 1) * 15/14), so source column s is read (with a weight that is
// not zero) by the destination columns floor(14 s / 15) ... ceil(14 (s + 1) / 15) - 1, and likewise 21/23 for the rows
// -- checked for every source column and row against the tap tables in sf_create.  (Round 1's bound was a column and
// two rows wider: a missile's 8 x 9 destination pixels, two rounds of lanes, are really 7 x 7.)
// ... and the other way round: a destination pixel reads two adjacent source columns and up to three adjacent rows, so the
// destination pixels that read `b` read nothing further than this outside it.  A picture saved with its 84x84 part is
// good where nothing else is drawn within that reach of its box (kReachX, kReachY).
// (kReachX, kReachY: sf_drawrec.h)
constexpr int kTapColPeriod = 14, kTapRowPeriod = 21;  // destination columns / rows after which the INTER_AREA taps repeat
__device__ __forceinline__ Box out_box(const Box& b) {
  Box o;
  o.x0 = max((b.x0 * 14) / 15, 0);
  o.x1 = min((b.x1 * 14 + 14) / 15, SF_OUT);
  o.y0 = max((b.y0 * 21) / 23, 0);
  o.y1 = min((b.y1 * 21 + 22) / 23, SF_OUT);
  return o;
}

// The frame of one env: the 90x92 surface in LDS and (RESIZE) its 84x84 INTER_AREA image -- which lives where it
// is going anyway, in the caller's frame in HBM: it starts as a copy of the resampled background and only the few
// dozen pixels an object touches are rewritten (byte stores that merge in L2).  Keeping it in LDS too cost 7 of
// the 18 KB per env, i.e. waves per CU, on a kernel that needs them to hide its LDS round trips.
// inclusive prefix sum across the wave: six DPP adds (row_shr 1, 2, 4, 8; row_bcast 15 into rows 1 and 3, 31 into rows 2 and 3)
__device__ __forceinline__ int wave_inclusive_sum(int v) {
#define SF_SCAN_STEP(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, (ctrl), (rmask), 0xf, false)
  SF_SCAN_STEP(0x111, 0xf);
  SF_SCAN_STEP(0x112, 0xf);
  SF_SCAN_STEP(0x114, 0xf);
  SF_SCAN_STEP(0x118, 0xf);
  SF_SCAN_STEP(0x142, 0xa);
  SF_SCAN_STEP(0x143, 0xc);
#undef SF_SCAN_STEP
  return v;
}
template <bool RESIZE>
struct Frame {
  uint8_t* fb;
  uint8_t* obuf;        // 84 x 84, row stride SF_OUT (global memory; LDS in sf_fort_patch_kernel)
  const uint32_t* tab;  // the tap tables (sf_raster.h): global memory, or an LDS copy (the picture kernels)
  const uint32_t* ptab; // LDS: one PERIOD of the tap tables, or null (then `tab` is read): see resample_into
  int lane;
  uint32_t* tor;        // LDS: sf_tor_dev.h's records, objects and accumulators (kTorWords); the resample pass's records after them
  // One PERIOD of the tap tables for resample_quad, in LDS behind the resample pass's own records (the strokes' records are
  // dead by then): 90 / 84 = 15 / 14 and 92 / 84 = 23 / 21, so the column taps repeat every 14 destination columns -- every 28
  // = 7 quads of four -- with the first source column moving on by 30, the row taps every 21 rows with the first source row
  // moving on by 23 (checked entry by entry in sf_create).  kLtabEntries entries of 16 bytes: columns 0 .. 27, then rows
  // 0 .. 20.  A lane holds its entry in registers from the prologue on (lt_e: ONE 16-byte load per lane and frame) and the
  // wave drops the 784 bytes into LDS whenever a resample pass is about to run (fill_ltab) -- where every round of
  // resample_quad used to pull five table entries per lane, 5 KB per round, through the vector cache: the frame kernel's L1
  // moves 33 KB per frame, and what two more 16-byte loads per lane cost it was measured (+1.2 us per 16 384 frames).
  uint32_t* ltab;       // or null: the tables are read where `tab` points (the picture kernels' LDS copy)
  uint4 lt_e;
  static constexpr int kLtabCols = 28, kLtabEntries = kLtabCols + 21;
  __device__ __forceinline__ void fill_ltab() const {
    if (ltab) {  // (uniform, known at compile time per kernel)
      if (lane < kLtabEntries) reinterpret_cast<uint4*>(ltab)[lane] = lt_e;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }

  // cv2.resize(..., INTER_AREA) restricted to the destination pixels that read source pixels of `b`
  // (OpenCV's resizeArea_ arithmetic: per source row buf = sum alpha * S, then sum += beta * buf in
  // table order, saturate_cast<uchar>).
  __device__ __forceinline__ void resample(const Box& b) const {
    if (!SF_RESAMPLE_QUADS) {
      resample_into(b, obuf, SF_OUT, 0, 0);
      return;
    }
    if (!RESIZE || (SF_RENDER_SKIP & 16) || b.empty()) return;
    fill_ltab();
    // in groups of four pixels of a row (resample_quad): the box's columns widened to multiples of four
    const Box o = out_box(b);
    const int gx0 = o.x0 >> 2, gw = ((o.x1 + 3) >> 2) - gx0, n = gw * (o.y1 - o.y0);
    const float r_gw = recip_i(gw);
    for (int i = lane; i < n; i += 64) {
      const DivMod dm = fast_divmod(i, gw, r_gw);
      resample_quad(4 * (gx0 + dm.r), o.y0 + dm.q);
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ... written to dst[(dy - y_off) * stride + (dx - x_off)]
  // one destination pixel (dx, dy) of the 84x84 image from the surface as it is (OpenCV's resizeArea_ arithmetic): the taps, then the sums
  struct Taps {
    uint4 c, r;  // column: first source column, a0, a1, -; row: first source row, b0, b1, b2
  };
  __device__ __forceinline__ Taps taps_fetch(int dx, int dy) const {
    Taps t;
    if (ptab) {  // uniform
      // 90 / 84 = 15 / 14 and 92 / 84 = 23 / 21: the taps repeat exactly every 14 columns / 21 rows (checked entry by
      // entry in sf_create), the first source cell moving on by 15 / 23.  One period sits in LDS.
      const int qx = (dx * 37) >> 9, px_ = dx - 14 * qx;   // dx / 14, dx % 14 for dx < 84
      const int qy = (dy * 49) >> 10, py_ = dy - 21 * qy;  // dy / 21, dy % 21 for dy < 84
      t.c = *reinterpret_cast<const uint4*>(ptab + 4 * px_);
      t.r = *reinterpret_cast<const uint4*>(ptab + 4 * (kTapColPeriod + py_));
      t.c.x += 15u * (unsigned)qx;
      t.r.x += 23u * (unsigned)qy;
    } else {
      // (unsigned indices: base pointer in scalar registers + a 32-bit offset per lane, no 64-bit address arithmetic)
      t.c = *reinterpret_cast<const uint4*>(tab + 4u * (unsigned)dx);
      t.r = *reinterpret_cast<const uint4*>(tab + 4u * (unsigned)(SF_OUT + dy));
    }
    return t;
  }
  __device__ __forceinline__ void resample_finish(const Taps& t, int dx, int dy, uint8_t* dst, int stride, int x_off, int y_off) const {
    const int fx = (int)t.c.x, fy = (int)t.r.x;
    const float a0 = __uint_as_float(t.c.y), a1 = __uint_as_float(t.c.z);
    const float b0 = __uint_as_float(t.r.y), b1 = __uint_as_float(t.r.z), b2 = __uint_as_float(t.r.w);
    const uint8_t* r0 = fb + fy * SF_IMG_W + fx;
    const uint8_t* r1 = r0 + SF_IMG_W;
    const uint8_t* r2 = r1 + SF_IMG_W;
    const float h0 = (float)r0[0] * a0 + (float)r0[1] * a1;
    const float h1 = (float)r1[0] * a0 + (float)r1[1] * a1;
    const float h2 = (float)r2[0] * a0 + (float)r2[1] * a1;
    const float sum = (b0 * h0 + b1 * h1) + b2 * h2;  // a two-row entry has b2 = 0: adds +0
    int v = (int)rintf(sum);                            // saturate_cast<uchar>: round half to even, clamp
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    if (!(SF_RENDER_SKIP & 16384) || v == 77) dst[(unsigned)((dy - y_off) * stride + (dx - x_off))] = (uint8_t)v;  // (16384: timing only)
  }
  __device__ __forceinline__ void resample_px(int dx, int dy, uint8_t* dst, int stride, int x_off, int y_off) const {
    resample_finish(taps_fetch(dx, dy), dx, dy, dst, stride, x_off, y_off);
  }
  // FOUR destination pixels (dx0 .. dx0 + 3, dx0 a multiple of 4, row dy) of the 84x84 image, one aligned 32-bit store: the
  // per-pixel arithmetic of resample_px on the same operands -- every pixel the same byte -- at half the instructions a pixel:
  // one row entry, three 8-byte reads of the surface (four destination columns read at most six source columns) instead of
  // twelve 2-byte ones, one store instead of four.  A pixel beside an object's box that is swept along gets the value it has:
  // the image is the resampled surface everywhere, at all times before the score's / bar's pictures go in.
  __device__ __forceinline__ void resample_quad(int dx0, int dy) const {
    uint4 tr, c0, c1, c2, c3;
    int first = 0;  // what the period's entries leave out of the first source cell: 23 rows per 21, 30 columns per 28
    if (ltab) {  // (uniform)
      const uint4* lt = reinterpret_cast<const uint4*>(ltab);
      const unsigned q = (unsigned)dx0 >> 2, qq = (q * 37u) >> 8, q7 = q - 7u * qq;  // q / 7, q % 7 for q < 21
      const unsigned qy = ((unsigned)dy * 49u) >> 10, ry = (unsigned)dy - 21u * qy;  // dy / 21, dy % 21 for dy < 84
      tr = lt[kLtabCols + ry];
      c0 = lt[4u * q7];
      c1 = lt[4u * q7 + 1u];
      c2 = lt[4u * q7 + 2u];
      c3 = lt[4u * q7 + 3u];
      first = (int)(23u * qy) * SF_IMG_W + (int)(30u * qq);
    } else {
      const uint4* t4 = reinterpret_cast<const uint4*>(tab);
      tr = t4[(unsigned)(SF_OUT + dy)];
      c0 = t4[(unsigned)dx0];
      c1 = t4[(unsigned)dx0 + 1u];
      c2 = t4[(unsigned)dx0 + 2u];
      c3 = t4[(unsigned)dx0 + 3u];
    }
    const float b0 = __uint_as_float(tr.y), b1 = __uint_as_float(tr.z), b2 = __uint_as_float(tr.w);
    const uint8_t* r0 = fb + first + (int)tr.x * SF_IMG_W + (int)c0.x;
    unsigned long long w0, w1, w2;
#ifndef SF_RQ_ALIGNED
#define SF_RQ_ALIGNED 0 /* A/B: a row's eight bytes as three ALIGNED words + two v_alignbyte instead of one 8-byte read at any byte */
#endif
#if SF_RQ_ALIGNED
    {
      static_assert(SF_IMG_W % 4 == 2, "the rows' alignments below");
      const unsigned A = (unsigned)(first + (int)tr.x * SF_IMG_W + (int)c0.x), al = A & 3u, al1 = (A + 2u) & 3u;  // (fb is 16-byte aligned)
      const uint32_t* q0 = reinterpret_cast<const uint32_t*>(fb + (A & ~3u));
      const uint32_t* q1 = reinterpret_cast<const uint32_t*>(fb + ((A + SF_IMG_W) & ~3u));
      const uint32_t* q2 = reinterpret_cast<const uint32_t*>(fb + ((A + 2 * SF_IMG_W) & ~3u));
      const unsigned a0 = q0[0], a1 = q0[1], a2 = q0[2], b0_ = q1[0], b1_ = q1[1], b2_ = q1[2], c0_ = q2[0], c1_ = q2[1], c2_ = q2[2];
      w0 = (unsigned long long)__builtin_amdgcn_alignbyte(a1, a0, al) | ((unsigned long long)__builtin_amdgcn_alignbyte(a2, a1, al) << 32);
      w1 = (unsigned long long)__builtin_amdgcn_alignbyte(b1_, b0_, al1) | ((unsigned long long)__builtin_amdgcn_alignbyte(b2_, b1_, al1) << 32);
      w2 = (unsigned long long)__builtin_amdgcn_alignbyte(c1_, c0_, al) | ((unsigned long long)__builtin_amdgcn_alignbyte(c2_, c1_, al) << 32);
    }
#else
    __builtin_memcpy(&w0, r0, 8);
    __builtin_memcpy(&w1, r0 + SF_IMG_W, 8);
    __builtin_memcpy(&w2, r0 + 2 * SF_IMG_W, 8);
#endif
#ifndef SF_RQ_PERM
#define SF_RQ_PERM 1 /* A/B: the pixel's two source bytes by v_perm_b32 (one selector for the three rows) instead of three 64-bit
                        shifts, the rounded, saturated byte dropped into the word by v_cvt_pk_u8_f32 (round to nearest even,
                        clamp to 0 .. 255: saturate_cast<uchar>(rint)) instead of rint + clamp + shift + or */
#endif
#if SF_RQ_PERM
    unsigned word = 0u;
    auto one = [&](const uint4& c, unsigned k) {
      // bytes off, off + 1 of the row's eight (off = 0 .. 4), the upper half zero (selector byte 0x0c)
      const unsigned sel = (c.x - c0.x) * 0x0101u + 0x0c0c0100u;
      const unsigned p0 = __builtin_amdgcn_perm((unsigned)(w0 >> 32), (unsigned)w0, sel);
      const unsigned p1 = __builtin_amdgcn_perm((unsigned)(w1 >> 32), (unsigned)w1, sel);
      const unsigned p2 = __builtin_amdgcn_perm((unsigned)(w2 >> 32), (unsigned)w2, sel);
      const float a0 = __uint_as_float(c.y), a1 = __uint_as_float(c.z);
      const float h0 = (float)(p0 & 255u) * a0 + (float)(p0 >> 8) * a1;
      const float h1 = (float)(p1 & 255u) * a0 + (float)(p1 >> 8) * a1;
      const float h2 = (float)(p2 & 255u) * a0 + (float)(p2 >> 8) * a1;
      const float sum = (b0 * h0 + b1 * h1) + b2 * h2;
      word = __builtin_amdgcn_cvt_pk_u8_f32(sum, k, word);
    };
    one(c0, 0u);
    one(c1, 1u);
    one(c2, 2u);
    one(c3, 3u);
#else
    auto one = [&](const uint4& c) -> unsigned {
      const unsigned sh = 8u * (c.x - c0.x);  // 0 .. 32: the pixel's two source columns are bytes sh / 8 and sh / 8 + 1
      const unsigned p0 = (unsigned)(w0 >> sh), p1 = (unsigned)(w1 >> sh), p2 = (unsigned)(w2 >> sh);
      const float a0 = __uint_as_float(c.y), a1 = __uint_as_float(c.z);
      const float h0 = (float)(p0 & 255u) * a0 + (float)((p0 >> 8) & 255u) * a1;
      const float h1 = (float)(p1 & 255u) * a0 + (float)((p1 >> 8) & 255u) * a1;
      const float h2 = (float)(p2 & 255u) * a0 + (float)((p2 >> 8) & 255u) * a1;
      const float sum = (b0 * h0 + b1 * h1) + b2 * h2;
      int v = (int)rintf(sum);
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      return (unsigned)v;
    };
    const unsigned word = one(c0) | (one(c1) << 8) | (one(c2) << 16) | (one(c3) << 24);
#endif
    *reinterpret_cast<uint32_t*>(obuf + (unsigned)(dy * SF_OUT + dx0)) = word;
  }
  __device__ __forceinline__ void resample_into(const Box& b, uint8_t* dst, int stride, int x_off, int y_off) const {
    if (!RESIZE || (SF_RENDER_SKIP & 16) || b.empty()) return;
    const Box o = out_box(b);
    const int ox0 = o.x0, oy0 = o.y0;
    const int ow = o.x1 - o.x0, n = ow * (o.y1 - o.y0);
    const float r_ow = recip_i(ow);
    for (int i = lane; i < n; i += 64) {
      const DivMod dm = fast_divmod(i, ow, r_ow);
      resample_px(ox0 + dm.r, oy0 + dm.q, dst, stride, x_off, y_off);
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- Small objects: the ship (3 lines), missiles (3 each), shells (4 each): each object ONE cairo_stroke of the reference
  // (drawWireFrame, SRC/draw.cpp:82-100), i.e. the union of its lines' rectangles through cairo's scan conversion.  Lane s holds
  // line s of the frame's draw order as a fixed-point quad (`mine`, `valid`); `obj0` = the first lane of the line's object, `kind`
  // the object's (sf_tor_dev.h).  sftd::raster takes up to sixteen lines of whole objects at a time.
  static constexpr int kChunk = sftd::kMaxQuadsF;
  static_assert(SF_IMG_H * sft::kGridY < 4096 && SF_IMG_W * 256 + 65536 < (1 << 18) && SF_IMG_H <= 96, "sf_tor_dev.h's fast arrangement: packed sub-rows, cells, rows");
  static constexpr int kMapBitsOut = 1024;  // the map of the objects' 84x84 boxes in the resample pass
  static constexpr int kLtabAt = 4 * kChunk + kMapBitsOut / 32;  // words into `tor`: behind that pass's records and map
  static constexpr int kResampleWords = kLtabAt + 4 * kLtabEntries;
  static constexpr int kTorWords = sftd::kLdsWordsF;
  static_assert(kTorWords >= kResampleWords, "the resample pass's records fit");
  static_assert(kLtabAt % 4 == 0, "the period of the tap tables is 16-byte aligned");
  __device__ __forceinline__ sftd::CtxF tor_ctx_fast() const { return sftd::CtxF{tor, fb, SF_IMG_W, SF_IMG_H, lane}; }
  // every stroke of the frame goes through sf_tor_dev.h's fast arrangement: objects over the surface's left / right border
  // and curve pieces with three edges on a side included
  __device__ __forceinline__ void raster_any(const sft::Quad& q, bool valid, int obj0, int kind, int grey) const {
    sftd::raster_fast(tor_ctx_fast(), q, valid, obj0, kind, grey);
  }
  __device__ __forceinline__ void draw_strokes(const sft::Quad& mine, bool valid, int obj0, int kind) const {
#define SF_DS_STAMP(k) do { if (SF_RENDER_STOP == 41 + (k)) return; } while (0)
    const Box myb = quad_box(mine);
    const bool mineok = valid && !myb.empty();
    // an object whose box is empty in one of its lines may still touch the surface with another: keep whole objects -- but an
    // object NONE of whose lines touches the surface (a missile between the view's border and the game area's, a third of
    // its life) draws nothing and is left out: no record, no sub-rows, and a frame with nothing else makes no call at all
    const unsigned long long drawn = __ballot(mineok);
    const int nq = kind == sftd::kKindShell ? 4 : 3;  // (per lane: the object's)
#ifndef SF_CULL_OFFSCREEN
#define SF_CULL_OFFSCREEN 1
#endif
    const bool obj_seen = !SF_CULL_OFFSCREEN || ((drawn >> obj0) & ((1ull << nq) - 1ull)) != 0ull;
    unsigned long long live = __ballot(valid && obj_seen);
    SF_DS_STAMP(0);
    while (live) {
      // this chunk: the first objects whose lines fit sixteen records
      const int rank0 = (int)__popcll(live & ((1ull << obj0) - 1ull));
      const bool in = ((live >> lane) & 1ull) && rank0 + nq <= kChunk;
      const unsigned long long chunk = __ballot(in);
      live &= ~chunk;
      raster_any(mine, in, obj0, kind, 255);
      SF_DS_STAMP(2);
    }
    // ---- the 84x84 pixels that read what was drawn.  An object = the (at most four, consecutive) strokes that share
    // `obj0`; its box = the union of their boxes, gathered at its first lane with three whole-wave DPP shifts.  Then ONE
    // enumeration over the destination pixels of all the objects' boxes: prefix sum, starts in LDS, a lane per pixel finds its
    // object --, everything being drawn by now (a destination pixel that reads a changed source pixel lies in some object's box).
    SF_DS_STAMP(3);
    if (RESIZE && !(SF_RENDER_SKIP & 16) && drawn) {
      fill_ltab();  // (behind the objects' records and map of this pass: nothing below writes there)
      const bool me = (drawn >> lane) & 1ull;
      int ux0 = me ? myb.x0 : (1 << 20), uy0 = me ? myb.y0 : (1 << 20), ux1 = me ? myb.x1 : -1, uy1 = me ? myb.y1 : -1;
      // (gathered at the object's first lane WITH a box: its first line's may be empty)
      const unsigned long long same_before = drawn & ((1ull << lane) - 1ull) & ~((1ull << obj0) - 1ull);
      const bool head = me && same_before == 0ull;
      {
        int sx0 = ux0, sy0 = uy0, sx1 = ux1, sy1 = uy1, so = obj0;
#pragma unroll
        for (int d = 1; d < 4; d++) {  // lane l sees lane l + d: wave_shl:1, applied d times; lanes past the end read the identity
          sx0 = __builtin_amdgcn_update_dpp(1 << 20, sx0, 0x130, 0xf, 0xf, false);
          sy0 = __builtin_amdgcn_update_dpp(1 << 20, sy0, 0x130, 0xf, 0xf, false);
          sx1 = __builtin_amdgcn_update_dpp(-1, sx1, 0x130, 0xf, 0xf, false);
          sy1 = __builtin_amdgcn_update_dpp(-1, sy1, 0x130, 0xf, 0xf, false);
          so = __builtin_amdgcn_update_dpp(-1, so, 0x130, 0xf, 0xf, false);
          const bool same = so == obj0;
          ux0 = same ? min(ux0, sx0) : ux0;
          uy0 = same ? min(uy0, sy0) : uy0;
          ux1 = same ? max(ux1, sx1) : ux1;
          uy1 = same ? max(uy1, sy1) : uy1;
        }
      }
      const Box o = out_box(Box{ux0, uy0, ux1, uy1});
      // (SF_RESAMPLE_QUADS: the unit of the enumeration is four pixels of a row, at multiples of four: resample_quad)
      const int ogx0 = SF_RESAMPLE_QUADS ? o.x0 >> 2 : o.x0;
      const int ow = SF_RESAMPLE_QUADS ? ((o.x1 + 3) >> 2) - ogx0 : o.x1 - o.x0;
      const int on = (head && ux1 > ux0 && uy1 > uy0) ? ow * (o.y1 - o.y0) : 0;
      unsigned long long todo = __ballot(on > 0);
      while (todo) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)todo, 0u));
        const bool cand = ((todo >> lane) & 1ull) && rank < kChunk;
        int incl = cand ? on : 0;
        const int mine_n = incl;
        incl = wave_inclusive_sum(incl);
        const int myoff = incl - mine_n;
        const bool own = cand && myoff <= kMapBitsOut;
        const unsigned long long chunk = __ballot(own);
        todo &= ~chunk;
        const int total = __builtin_amdgcn_readlane(incl, 63 - __builtin_clzll(chunk));
        int* const orec = reinterpret_cast<int*>(tor);  // per object: x0 | y0 << 8, w, start, 1 / w
        uint32_t* const omap = tor + 4 * kChunk;
        if (lane < kMapBitsOut / 32) omap[lane] = 0u;
        if (own) {
          if (myoff > 0) atomicOr(&omap[(myoff - 1) >> 5], 1u << ((myoff - 1) & 31));
          orec[4 * rank] = ogx0 | (o.y0 << 8);
          orec[4 * rank + 1] = ow;
          orec[4 * rank + 2] = myoff;
          reinterpret_cast<float*>(orec)[4 * rank + 3] = recip_i(ow);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int kb = 0;
        for (int base = 0; base < total; base += 64) {
          const int i = base + lane;
          const uint2 mw = base < kMapBitsOut ? *reinterpret_cast<const uint2*>(omap + (base >> 5)) : uint2{0u, 0u};
          const int k = kb + (int)__builtin_amdgcn_mbcnt_hi(mw.y, __builtin_amdgcn_mbcnt_lo(mw.x, 0u));
          kb += __popc(mw.x) + __popc(mw.y);
          if (i < total) {
            const int4 rec = *reinterpret_cast<const int4*>(orec + 4 * k);
            const DivMod dm = fast_divmod(i - rec.z, rec.y, __int_as_float(rec.w));
            if (SF_RESAMPLE_QUADS) resample_quad(4 * ((rec.x & 255) + dm.r), (rec.x >> 8) + dm.q);
            else resample_px((rec.x & 255) + dm.r, (rec.x >> 8) + dm.q, obuf, SF_OUT, 0, 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      // (the accumulators behind the records were zero and must be again: the pass above wrote over the records only --
      //  kResampleWords <= the records' and objects' space)
    }
#undef SF_DS_STAMP
  }
};
static_assert(Frame<true>::kResampleWords <= sftd::kMaxQuads * sftd::kRecWords + sftd::kMaxObjs * sftd::kObjWords &&
              Frame<true>::kResampleWords <= sftd::kMaxQuadsF * sftd::kRecWordsF + sftd::kMaxObjs * sftd::kObjWordsF,
              "the resample pass's records, map and tap period stay clear of sf_tor_dev.h's accumulators (which must stay zero)");

// ---- the objects' lines as cairo has them: path points through the matrices of drawGameStateScaled + drawWireFrame
// (sf_tor.h: view_matrix, object_matrix, to_device), in float64 like cairo, then one butt-capped rectangle per line
// wireframe segments (ax, ay, bx, by), SRC/wireframe.cpp:11-67, as functions of the line's index (small whole numbers)
struct Seg4 {
  double ax, ay, bx, by;
};
__device__ __forceinline__ Seg4 ship_seg(int k) {      // {-18, 0, 18, 0}, {-18, 18, 0, 0}, {0, 0, -18, -18}
  return Seg4{k == 2 ? 0.0 : -18.0, k == 1 ? 18.0 : 0.0, k == 0 ? 18.0 : (k == 1 ? 0.0 : -18.0), k == 2 ? -18.0 : 0.0};
}
__device__ __forceinline__ Seg4 missile_seg(int k) {   // {0, 0, -25, 0}, {0, 0, -5, 5}, {0, 0, -5, -5}
  return Seg4{0.0, 0.0, k == 0 ? -25.0 : -5.0, k == 0 ? 0.0 : (k == 1 ? 5.0 : -5.0)};
}
__device__ __forceinline__ Seg4 shell_seg(int k) {     // {-8, 0, 0, -6}, {0, -6, 16, 0}, {16, 0, 0, 6}, {0, 6, -8, 0}
  return Seg4{k == 0 ? -8.0 : (k == 2 ? 16.0 : 0.0), k == 1 ? -6.0 : (k == 3 ? 6.0 : 0.0), k == 1 ? 16.0 : (k == 3 ? -8.0 : 0.0),
              k == 0 ? -6.0 : (k == 2 ? 6.0 : 0.0)};
}
__device__ __forceinline__ sft::Affine default_view() { return sft::view_matrix(SF_SCALE, SF_SCALE, SF_VP_X, SF_VP_Y); }
// one line of a wireframe at (px, py), heading `deg` (cos / sin of deg2rad(deg) from the batch's table: the host's libm)
__device__ __forceinline__ sft::Quad line_quad(const Seg4& g, double px, double py, const double* trig, int deg) {
  deg = deg < 0 ? 0 : (deg > 359 ? 359 : deg);  // (never out of range for a state the step kernel produced)
  const double2 cs = *reinterpret_cast<const double2*>(trig + 2 * deg);
  const sft::Affine m = sft::object_matrix(default_view(), px, py, cs.x, cs.y);
  int x1, y1, x2, y2;
  sft::to_device(m, g.ax, g.ay, &x1, &y1);
  sft::to_device(m, g.bx, g.by, &x2, &y2);
  return sft::stroke_quad(x1, y1, x2, y2, SF_SCALE, SF_SCALE, SF_LINE_W / 2);
}

// drawExplosion (SRC/draw.cpp:116-145): 7 rings (radius 15 + 8 i) of twelve 10-degree arcs starting at 30 k + 3 (i + 1)
// degrees, each its own cairo_stroke -- one quad between the faces at its two ends (sf_tor.h: arc_quad_fixed) --, then the
// radius-7 circle, one stroke of sixteen pieces.  `arcs` = the batch's table of the 84 + 2 arcs' constants (sft::ArcK, made
// by the host's libm like cairo makes them).  Ring by ring: twelve arcs at once, each its own object.
// call `ring` (0 .. 6: a ring's twelve arcs, each an object of its own; 7: the circle, one object of sixteen pieces) as the lanes'
// quads for the rasteriser
struct RingCall {
  sft::Quad q;
  bool valid;
  int obj0, kind, grey;
};
__device__ __forceinline__ RingCall explosion_call(int ring, int lane, const double* arcs, double cx, double cy) {
  const sft::Affine v = default_view();
  RingCall c;
  c.q = sft::Quad{};
  if (ring < 7) {
    c.valid = lane < 12;
    if (c.valid) {
      const double* kp = arcs + 8 * (12 * ring + lane);
      const sft::ArcK k{kp[0], kp[1], kp[2], kp[3], kp[4], kp[5], kp[6], kp[7]};
      c.q = sft::arc_quad_fixed(sft::arc_knots(v, cx, cy, k), SF_SCALE, SF_SCALE, (double)(float)SF_LINE_W / 2);
    }
    c.obj0 = lane;
    c.kind = sftd::kKindSingle;
    c.grey = 15 + 8 * ring < 60 ? 191 : 128;  // .75 / .5
  } else {
    c.valid = lane < 16;
    if (c.valid) {
      const double* kp = arcs + 8 * (84 + (lane >> 3));
      const sft::ArcK k{kp[0], kp[1], kp[2], kp[3], kp[4], kp[5], kp[6], kp[7]};
      c.q = sft::ring_piece_quad(sft::arc_knots(v, cx, cy, k), lane & 7, SF_SCALE, SF_SCALE, (double)(float)SF_LINE_W / 2);
    }
    c.obj0 = 0;
    c.kind = sftd::kKindRing | (8 << 8);
    c.grey = 191;
  }
  return c;
}
template <bool RESIZE>
__device__ __forceinline__ void draw_explosion(const Frame<RESIZE>& F, const double* arcs, double cx, double cy) {
  const int lane = F.lane;
#pragma unroll 1
  for (int ring = 0; ring < 8; ring++) {
    const RingCall c = explosion_call(ring, lane, arcs, cx, cy);
    F.raster_any(c.q, c.valid, c.obj0, c.kind, c.grey);
  }
  F.resample(explosion_box((float)cx, (float)cy));
}

// four bytes of a cached rectangle to p (any alignment; global memory and LDS both take an unaligned 32-bit store),
// or the first n of them when the rectangle ends inside the word
__device__ __forceinline__ void put_bytes(uint8_t* p, uint32_t w, int n) {
  if (n >= 4) {
    __builtin_memcpy(p, &w, 4);
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (k < n) p[k] = (uint8_t)(w >> (8 * k));
  }
}
// The same for a rectangle whose rows are held a word per LANE, consecutive lanes consecutive words of a row at least four
// bytes wide: the word that straddles the row's end (n = 1 .. 3 of its bytes belong to the row) is stored four bytes wide
// all the same, ENDING at the row's end -- its front filled with the tail of the previous word, which the lane below holds
// and has just written to those very bytes.  One store per lane and no branch on n, where put_bytes is a branch and up to
// three byte stores under three more: a cached explosion is seven such words per lane, every frame of a dead ship.
// (`lo` = the previous word of the row; n <= 0: the lane has nothing in this row.)
__device__ __forceinline__ uint32_t lane_below(uint32_t w) {  // lane l gets lane l - 1's value (wave_shr:1; lane 0: its own)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)w, (int)w, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ void put_row_word(uint8_t* p, uint32_t w, uint32_t lo, int n) {
  const uint32_t v = n >= 4 ? w : __builtin_amdgcn_alignbyte(w, lo, (unsigned)n);  // bytes n .. 3 of lo, then 0 .. n - 1 of w
  uint8_t* q = p + (n >= 4 ? 0 : n - 4);
  if (n > 0) __builtin_memcpy(q, &v, 4);
}

// ---- A dead ship stays where it died for the 1000 ms of its explosion (30 frames), and the
// explosion is the first thing drawn on the static background: its pixels -- and the 84x84 pixels
// that read them -- are a function of the position alone.  Each env keeps them in HBM, keyed by that
// position: SF_XC_BYTES per env = {x, y (f64 bits), flags, pad; 27 rows x 28 of the surface; 28 rows
// x 28 of the 84x84 image}.  29 of 30 explosion frames become two small copies.
constexpr int kXcKey = 0, kXcFlags = 16, kXcFb = 32, kXcRow = 28, kXcFbRows = 27, kXcOut = kXcFb + kXcFbRows * kXcRow + 12,
              kXcOutRows = 28;
// ... and, for an explosion that reaches the score or the bar (a ship lost through the upper or lower edge of the big
// hexagon: a sixth of the losses each), what their boxes end up as with the score / bar drawn over it: the points (int) and
// the bar's state (int) they were drawn for at kXcHudKeys, the two pictures (hud_picture's layout) behind.  Flags: 1, 2 =
// the explosion's surface / 84x84 part; 4, 8 = the score box's; 16, 32 = the bar box's.
constexpr int kXcHudKeys = 1584, kXcScore = 1600, kXcBar = kXcScore + SF_HUD_SCORE_BYTES;
static_assert(kXcOut % 4 == 0 && kXcOut + kXcOutRows * kXcRow <= kXcHudKeys && kXcBar + SF_HUD_BAR_BYTES <= SF_XC_BYTES,
              "explosion cache layout");
struct XcState {
  unsigned flags;  // of the entry as it is after ship_explosion (0: no usable entry)
  int points, bar; // the keys of its score / bar pictures
};

template <bool RESIZE>
__device__ __forceinline__ XcState ship_explosion(const Frame<RESIZE>& F, const double* arcs, unsigned char* xc, double x, double y,
                                                  const bool fill = true, const bool skip_lookup = false) {
  const float cx = (float)x, cy = (float)y;
  const Box b = explosion_box(cx, cy), o = out_box(b);
  const int lane = F.lane;
  const unsigned need = RESIZE ? 3u : 1u;
  bool hit = false;
  XcState st{0u, 0, 0};
  if (xc && !skip_lookup) {  // (skip_lookup: the caller has looked already -- xc_fetch / xc_apply -- and it was a miss)
    const double kx = *reinterpret_cast<const double*>(xc + kXcKey), ky = *reinterpret_cast<const double*>(xc + kXcKey + 8);
    const unsigned fl = *reinterpret_cast<const unsigned*>(xc + kXcFlags);
    const int2 keys = *reinterpret_cast<const int2*>(xc + kXcHudKeys);
    hit = (kx == x && ky == y && (fl & need) == need) || (SF_RENDER_SKIP & 1024);  // (bit 10: a miss costs what a hit does)
    st = XcState{fl, keys.x, keys.y};
  }
  // (put_row_word: rows of sixteen bytes or more, see kXcRowW -- a ship dies inside the big hexagon, 5 pixels or more from the
  //  surface's edges: its explosion's box is never cut below 17)
  const bool fits = b.x1 - b.x0 <= kXcRow && b.y1 - b.y0 <= kXcFbRows && o.x1 - o.x0 <= kXcRow && o.y1 - o.y0 <= kXcOutRows &&
                    b.x1 - b.x0 >= 16 && o.x1 - o.x0 >= 16;
  if (hit && fits) {
    // every load first (3 + 4 dwords per lane), then the byte writes: one memory round trip instead of 25
    constexpr int kRowW = kXcRow / 4;  // 7 dwords a row
    const uint32_t* gf = reinterpret_cast<const uint32_t*>(xc + kXcFb);
    const uint32_t* go = reinterpret_cast<const uint32_t*>(xc + kXcOut);
    uint32_t wf[3], wo[4];
#pragma unroll
    for (int j = 0; j < 3; j++) wf[j] = (lane + 64 * j < kXcFbRows * kRowW) ? gf[lane + 64 * j] : 0u;
    if (RESIZE) {
#pragma unroll
      for (int j = 0; j < 4; j++) wo[j] = (lane + 64 * j < kXcOutRows * kRowW) ? go[lane + 64 * j] : 0u;
    }
    const int bw = b.x1 - b.x0, bh = b.y1 - b.y0, ow = o.x1 - o.x0, oh = o.y1 - o.y0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int d = lane + 64 * j, r = d / kRowW, c4 = (d - r * kRowW) * 4;
      put_row_word(F.fb + (b.y0 + r) * SF_IMG_W + b.x0 + c4, wf[j], lane_below(wf[j]), r < bh ? bw - c4 : 0);
    }
    if (RESIZE) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int d = lane + 64 * j, r = d / kRowW, c4 = (d - r * kRowW) * 4;
        put_row_word(F.obuf + (o.y0 + r) * SF_OUT + o.x0 + c4, wo[j], lane_below(wo[j]), r < oh ? ow - c4 : 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    return st;
  }
  draw_explosion(F, arcs, x, y);
  st.flags = 0;
  if (xc && fits && fill) {
    for (int i = lane; i < kXcFbRows * kXcRow; i += 64) {
      const int r = i / kXcRow, c = i - r * kXcRow;
      if (r < b.y1 - b.y0 && c < b.x1 - b.x0) xc[kXcFb + i] = F.fb[(b.y0 + r) * SF_IMG_W + b.x0 + c];
    }
    F.resample_into(b, xc + kXcOut, kXcRow, o.x0, o.y0);  // the same values draw_explosion just gave the frame
    if (lane == 0) {
      *reinterpret_cast<double*>(xc + kXcKey) = x;
      *reinterpret_cast<double*>(xc + kXcKey + 8) = y;
      *reinterpret_cast<unsigned*>(xc + kXcFlags) = need;
    }
    st.flags = need;  // (a new explosion: whatever score / bar picture the entry held belongs to the old one)
  }
  return st;
}

// ship_explosion's lookup in two halves, for the frame kernel: the entry's key AND its pixels are asked for in the
// prologue's one round trip (xc_fetch: 16 + 12 + 28 bytes per lane, whether or not the key will match), and used right
// behind the surface's arrival (xc_apply) -- instead of a round trip for the key and, on a hit, another for the pixels.
struct XcFetch {
  double kx, ky;
  unsigned fl;
  int2 keys;
  uint32_t wf[3], wo[4];
};
constexpr int kXcRowW = kXcRow / 4;  // 7 dwords a row
// (put_row_word takes the previous word of a row from the lane below.  Lane 0's word is dword 64 j of the rectangle, column
//  4 (64 j % 7) = 4, 8, 12 for j = 1, 2, 3: it is a full word -- never a row's cut last one -- in every row of 16 bytes or
//  more, which is what `fits` lets through)
static_assert((64 % kXcRowW) * 4 + 4 <= 16 && (128 % kXcRowW) * 4 + 4 <= 16 && (192 % kXcRowW) * 4 + 4 <= 16, "lane 0 of a later round");
template <bool RESIZE>
__device__ __forceinline__ XcFetch xc_fetch(const unsigned char* xc, int lane) {
  XcFetch f;
  f.kx = *reinterpret_cast<const double*>(xc + kXcKey);
  f.ky = *reinterpret_cast<const double*>(xc + kXcKey + 8);
  f.fl = *reinterpret_cast<const unsigned*>(xc + kXcFlags);
  f.keys = *reinterpret_cast<const int2*>(xc + kXcHudKeys);
  const uint32_t* gf = reinterpret_cast<const uint32_t*>(xc + kXcFb);
  const uint32_t* go = reinterpret_cast<const uint32_t*>(xc + kXcOut);
#pragma unroll
  for (int j = 0; j < 3; j++) f.wf[j] = (lane + 64 * j < kXcFbRows * kXcRowW) ? gf[lane + 64 * j] : 0u;
#pragma unroll
  for (int j = 0; j < 4; j++) f.wo[j] = (RESIZE && lane + 64 * j < kXcOutRows * kXcRowW) ? go[lane + 64 * j] : 0u;
  return f;
}
// true: the entry was this explosion's and its pixels are in the frame (*st = the entry's flags and keys); false: draw it
template <bool RESIZE>
__device__ __forceinline__ bool xc_apply(const Frame<RESIZE>& F, const XcFetch& f, double x, double y, XcState* st) {
  const float cx = (float)x, cy = (float)y;
  const Box b = explosion_box(cx, cy), o = out_box(b);
  const int lane = F.lane;
  const unsigned need = RESIZE ? 3u : 1u;
  const bool hit = (f.kx == x && f.ky == y && (f.fl & need) == need) || (SF_RENDER_SKIP & 1024);
  const bool fits = b.x1 - b.x0 <= kXcRow && b.y1 - b.y0 <= kXcFbRows && o.x1 - o.x0 <= kXcRow && o.y1 - o.y0 <= kXcOutRows &&
                    b.x1 - b.x0 >= 16 && o.x1 - o.x0 >= 16;
  if (!(hit && fits)) return false;
  const int bw = b.x1 - b.x0, bh = b.y1 - b.y0, ow = o.x1 - o.x0, oh = o.y1 - o.y0;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int d = lane + 64 * j, r = d / kXcRowW, c4 = (d - r * kXcRowW) * 4;
    put_row_word(F.fb + (b.y0 + r) * SF_IMG_W + b.x0 + c4, f.wf[j], lane_below(f.wf[j]), r < bh ? bw - c4 : 0);
  }
  if (RESIZE) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int d = lane + 64 * j, r = d / kXcRowW, c4 = (d - r * kXcRowW) * 4;
      put_row_word(F.obuf + (o.y0 + r) * SF_OUT + o.x0 + c4, f.wo[j], lane_below(f.wo[j]), r < oh ? ow - c4 : 0);
    }
  }
  __builtin_amdgcn_wave_barrier();
  *st = XcState{f.fl, f.keys.x, f.keys.y};
  return true;
}

}  // namespace

// Up to 7 x 64 pieces of 16 bytes (an 84x84 frame is 441) from src to dst, global memory both: every load in flight
// before the first store.  (Left as a loop, the compiler waits for each piece before it asks for the next -- seven L2
// round trips one after the other; with the pieces in an array it parks them in scratch.  Hence seven named registers,
// loads clamped into the range instead of predicated.)
constexpr int kFrameRounds = (kOutBytes / 16 + 63) / 64;
static_assert(kFrameRounds == 7, "copy_pieces is written out for seven rounds");
struct Pieces {
  uint4 v0, v1, v2, v3, v4, v5, v6;
};
__device__ __forceinline__ Pieces load_pieces(const uint4* src, int n, int lane) {
  const int last = n - 1;
  return Pieces{src[min(lane, last)],       src[min(lane + 64, last)],  src[min(lane + 128, last)], src[min(lane + 192, last)],
                src[min(lane + 256, last)], src[min(lane + 320, last)], src[min(lane + 384, last)]};
}
#ifndef SF_FRAME_NT
#define SF_FRAME_NT 1 /* the 84x84 background's stores leave non-temporal when the frame is a slot of a frame stack (out_stride >
                         a frame): the 4-frame ring of 16 384 envs is 462 MB that cycle through the 256 MB Infinity Cache and
                         push out what the step and the frame kernel read every step -- 52.3 -> 51.2 us per step into the
                         ring.  A flat output (115 MB, rewritten in place every step) stays resident and wants plain stores:
                         non-temporal there costs +1.1 us.  0: always plain; 2: always non-temporal (A/B) */
#endif
template <bool NT>
__device__ __forceinline__ void store_piece(uint4* p, const uint4& v) {
  if (NT) {
    typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u4v_t{v.x, v.y, v.z, v.w}, reinterpret_cast<u4v_t*>(p));
  } else {
    *p = v;
  }
}
// (the non-temporal form stores last piece first: written in the same order, the optimiser finds the two branches of the
//  caller's `if` starting with identical stores, hoists them in front of it and drops the hint they do not share)
template <bool NT = false>
__device__ __forceinline__ void store_pieces(const Pieces& p, uint4* dst, int n, int lane) {
  if (NT) {
    if (lane + 384 < n) store_piece<true>(dst + lane + 384, p.v6);
    if (lane + 320 < n) store_piece<true>(dst + lane + 320, p.v5);
    if (lane + 256 < n) store_piece<true>(dst + lane + 256, p.v4);
    if (lane + 192 < n) store_piece<true>(dst + lane + 192, p.v3);
    if (lane + 128 < n) store_piece<true>(dst + lane + 128, p.v2);
    if (lane + 64 < n) store_piece<true>(dst + lane + 64, p.v1);
    if (lane < n) store_piece<true>(dst + lane, p.v0);
    return;
  }
  if (lane < n) store_piece<false>(dst + lane, p.v0);
  if (lane + 64 < n) store_piece<false>(dst + lane + 64, p.v1);
  if (lane + 128 < n) store_piece<false>(dst + lane + 128, p.v2);
  if (lane + 192 < n) store_piece<false>(dst + lane + 192, p.v3);
  if (lane + 256 < n) store_piece<false>(dst + lane + 256, p.v4);
  if (lane + 320 < n) store_piece<false>(dst + lane + 320, p.v5);
  if (lane + 384 < n) store_piece<false>(dst + lane + 384, p.v6);
}
__device__ __forceinline__ void copy_pieces(const uint4* src, uint4* dst, int n, int lane) {
  store_pieces(load_pieces(src, n, lane), dst, n, lane);
}

struct SfRenderArgs {
  const unsigned char* state;   // the tiled state: the shells' positions and velocities (everything else comes from `draw`)
  const unsigned char* draw;    // the envs' draw records (sf_drawrec.h), SF_DR_BYTES each, current for `state`
  int n_envs;
  const uint32_t* bg;    // four variants of the static 92x90 background, SF_BG_STRIDE bytes apart:
                         // bit 0 = with the score 0000000, bit 1 = with the empty vulnerability bar
  const uint32_t* bg84;  // ... and their 84x84 INTER_AREA images, 7056 bytes apart
  const uint32_t* tabs;  // SF_TAB_WORDS, layout in sf_raster.h
  uint8_t* out;
  size_t out_stride;     // bytes from one env's frame to the next (>= the frame size, multiple of 16)
  unsigned char* xcache; // SF_XC_BYTES per env (zero-initialised), or null
  const unsigned char* fpatch;  // 36 x SF_FP_BYTES: the live fortress at 0, 10, ... 350 degrees on the bare background, or null
  // frame stack (sf_render_stack): `out` is slot stack_slot of stack_n frames per env; an env whose done flag is
  // set gets its other slots zeroed first (`current_obs *= masks`, rl/train.py:92-93); stack_done null = plain render
  const uint8_t* stack_done;
  int stack_slot, stack_n;
  // sf_render_shift: the stack of the PREVIOUS step (same layout); its frames 1 .. stack_n-1 become frames
  // 0 .. stack_n-2 of this one (zeros for a finished env), the new frame is rendered into the last slot: the
  // trainer's update_current_obs + rollouts.insert (rl/train.py:51-56,98) without a separate copy
  const uint8_t* stack_prev;
  // launch order (pick_env): per tile of 64 envs, the ones whose ship died in the last tick (written by the step
  // kernel), and the number of workgroups in front of the grid that draw such envs first; null / 0 = env = blockIdx
  const unsigned long long* hint;
  int n_front;
  const unsigned char* hud;  // SF_HUD_BYTES: the score / bar pictures (sf_hud_kernel), or null
  const double* trig;        // cos, sin of deg2rad(k), k = 0 .. 359, as the reference's libm gives them (sf_trig_table)
  const double* arcs;        // 86 x 8: the explosion's arcs (sft::ArcK; sf_arc_table)
  const unsigned char* falpha;  // 36 x 256: the live fortress's coverage over its 16 x 16 box, heading 10 k (sf_image_fort_alpha)
  const SfGlyphAtlas* glyphs;   // the score text's glyph atlas (sf_glyphs.h); null or gw == 0: the seven-segment fallback
};

// ---- Which env a workgroup draws.  The first frame of a dead ship's explosion costs about three ordinary frames (96
// strokes, then cached for the other 29); a launch is four rounds of equal waves, so an expensive wave that starts in
// the last round is the tail of the launch: 14 of 102 us at 16 384 envs.  Workgroups are dispatched in blockIdx order,
// hence: the grid is n_front + n_envs workgroups; workgroup i < n_front draws the i-th env of the hint words (and
// leaves if there are fewer), workgroup n_front + e draws env e unless one of the first n_front did.  The hint decides
// only WHEN a frame is drawn: any content of the words gives every env exactly one workgroup.
// Order of the hinted envs: by (tile & 63), then tile >> 6, then lane of the tile.  Returns -1 for "nothing to draw".
__device__ __forceinline__ int hinted_scan(const unsigned long long* hint, int n_tiles, int lane, int* mine) {
  // (four words in flight per lane and trip -- 16 384 envs are four words a lane: one round trip, where a plain loop was
  //  four, one behind the other --, and the scan by six DPP adds instead of six LDS permutes: every one of the n_front
  //  workgroups comes through here, most of them to find nothing to draw)
  int c = 0;
  for (int t0 = 0; t0 < n_tiles; t0 += 256) {
    unsigned long long w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int t = t0 + 64 * j + lane;
      w[j] = hint[t < n_tiles ? t : 0];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) c += (t0 + 64 * j + lane < n_tiles) ? __popcll(w[j]) : 0;
  }
  *mine = c;
  return wave_inclusive_sum(c);  // inclusive scan over the lanes
}

__device__ __forceinline__ int pick_env(const SfRenderArgs& a, int p, int lane) {
  const int n_tiles = (a.n_envs + 63) >> 6;
  if (p >= a.n_front) {
    const int e = p - a.n_front;
    const unsigned long long w = a.hint[e >> 6];  // uniform
    if (!((w >> (e & 63)) & 1ull)) return e;      // (nearly every workgroup: one scalar load)
    // a hinted env: drawn in front if its rank there is below n_front
    int mine;
    const int incl = hinted_scan(a.hint, n_tiles, lane, &mine);
    const int tile = e >> 6, tl = tile & 63;
    int rank = __builtin_amdgcn_readlane(incl - mine, tl) + __popcll(w & ((1ull << (e & 63)) - 1ull));
    for (int t = tl; t < tile; t += 64) rank += __popcll(a.hint[t]);
    return rank < a.n_front ? -1 : e;
  }
  int mine;
  const int incl = hinted_scan(a.hint, n_tiles, lane, &mine);
  if (p >= __builtin_amdgcn_readlane(incl, 63)) return -1;
  const int owner = __builtin_ctzll(__ballot(incl > p));
  int left = p - __builtin_amdgcn_readlane(incl - mine, owner);
  for (int t = owner; t < n_tiles; t += 64) {
    unsigned long long w = a.hint[t];  // uniform
    const int pc = __popcll(w);
    if (left < pc) {
      for (; left > 0; left--) w &= w - 1ull;
      const int e = t * 64 + __builtin_ctzll(w);
      return e < a.n_envs ? e : -1;  // (the step kernel never marks a lane behind the batch)
    }
    left -= pc;
  }
  return -1;
}

// The fortress never moves and its heading is a multiple of the 10-degree sector (SRC/game.cpp:205-208):
// 36 pictures.  sf_fort_patch_kernel draws them once per batch with the frame code below (so they are
// bit-identical to drawing in place); a frame whose ship / explosion pixels stay clear of the fortress's
// box then copies 256 + 306 bytes instead of rasterising four strokes.
constexpr int kFpOutRow = 20, kFpOutAt = 256;  // (the picture's box kFpX0 .. kFpY1: sf_drawrec.h)

template <bool RESIZE>
__device__ __forceinline__ void fort_patch_copy(const Frame<RESIZE>& F, unsigned char* gp, bool store) {
  const Box b{kFpX0, kFpY0, kFpX1, kFpY1}, o = out_box(b);
  const int lane = F.lane;
  if (!store) {
    // every load first (one dword of the surface patch, two of the image patch per lane), then the byte writes
    const uint32_t* g32 = reinterpret_cast<const uint32_t*>(gp);
    const int ow = o.x1 - o.x0, oh = o.y1 - o.y0;
    const uint32_t w = g32[lane];
    uint32_t v[2] = {0u, 0u};
    if (RESIZE) {
      v[0] = g32[kFpOutAt / 4 + lane];
      if (64 + lane < (kFpOutRow / 4) * oh) v[1] = g32[kFpOutAt / 4 + 64 + lane];
    }
    put_bytes(F.fb + (kFpY0 + (lane >> 2)) * SF_IMG_W + kFpX0 + (lane & 3) * 4, w, 4);
    if (RESIZE) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int d = lane + 64 * j, r = d / (kFpOutRow / 4), c4 = (d - r * (kFpOutRow / 4)) * 4;
        if (r < oh) put_bytes(F.obuf + (o.y0 + r) * SF_OUT + o.x0 + c4, v[j], ow - c4);
      }
    }
    __builtin_amdgcn_wave_barrier();
    return;
  }
  for (int i = lane; i < 256; i += 64) {
    uint8_t* p = F.fb + (kFpY0 + (i >> 4)) * SF_IMG_W + kFpX0 + (i & 15);
    if (store) gp[i] = *p; else *p = gp[i];
  }
  if (RESIZE) {
    const int ow = o.x1 - o.x0, oh = o.y1 - o.y0;  // 17 x 18
    for (int i = lane; i < kFpOutRow * oh; i += 64) {
      const int r = i / kFpOutRow, c = i - r * kFpOutRow;
      if (c < ow) {
        uint8_t* p = F.obuf + (o.y0 + r) * SF_OUT + o.x0 + c;
        if (store) gp[kFpOutAt + i] = *p; else *p = gp[kFpOutAt + i];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}


// the live fortress drawn IN PLACE (something the ship drew lies under it): its four lines are one cairo_stroke whose coverage
// depends on the heading alone -- an alpha map per sector, made by the host (sf_image_fort_alpha) --, white, lerped in
template <bool RESIZE>
__device__ __forceinline__ void fort_in_place(const Frame<RESIZE>& F, const unsigned char* falpha, int sector) {
  const uint32_t aw = reinterpret_cast<const uint32_t*>(falpha + 256 * sector)[F.lane];  // four pixels of a row of sixteen
  uint8_t* p = F.fb + (kFpY0 + (F.lane >> 2)) * SF_IMG_W + kFpX0 + (F.lane & 3) * 4;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int a = (int)((aw >> (8 * k)) & 255u);
    if (a) p[k] = (uint8_t)sft::lerp8(255, a, p[k]);
  }
  __builtin_amdgcn_wave_barrier();
  F.resample(Box{kFpX0, kFpY0, kFpX1, kFpY1});
}

// ---- score (drawScore, SRC/draw.cpp:161-173): "%07d", grey .5 through the glyph atlas's coverage (sf_glyphs.h), a lane per
// pixel of the box; without an atlas the seven-segment fallback (sf_raster.h)
template <bool RESIZE>
__device__ __forceinline__ void draw_score(const Frame<RESIZE>& F, int pnts, const SfGlyphAtlas* G) {
  constexpr int w = SF_TXT_BOX_X1 - SF_TXT_BOX_X0, h = SF_TXT_BOX_Y1 - SF_TXT_BOX_Y0;
  static_assert(SF_TXT_BOX_X0 <= (SF_TXT_X0 + SF_TXT_PAD - SF_VP_X) * SF_SCALE &&
                SF_TXT_BOX_X1 >= (SF_TXT_X0 + 6 * SF_TXT_ADV + SF_TXT_PAD + SF_TXT_W - SF_VP_X) * SF_SCALE &&
                SF_TXT_BOX_Y0 <= (SF_TXT_TOP - SF_VP_Y) * SF_SCALE &&
                SF_TXT_BOX_Y1 >= (SF_TXT_TOP + SF_TXT_H - SF_VP_Y) * SF_SCALE, "text box (fallback glyphs)");
  static_assert(SF_TXT_BOX_X0 <= sfg::kDefX0 && SF_TXT_BOX_X1 >= sfg::kDefX0 + 6 * sfg::kDefAdvance + sfg::kDefW &&
                SF_TXT_BOX_Y0 <= sfg::kDefY0 && SF_TXT_BOX_Y1 >= sfg::kDefY0 + sfg::kDefH, "text box (built-in atlas)");
  if (G && G->gw) {  // (uniform)
    const uint32_t chars = sfg::score_chars(pnts);
    for (int i = F.lane; i < w * h; i += 64) {
      const int ry = i / w, rx = i - ry * w;
      uint8_t* p = F.fb + (SF_TXT_BOX_Y0 + ry) * SF_IMG_W + SF_TXT_BOX_X0 + rx;
      *p = (uint8_t)sfg::text_pixel(G, chars, SF_TXT_BOX_X0 + rx, SF_TXT_BOX_Y0 + ry, *p);
    }
  } else {
    const unsigned long long masks = sfr::score_masks(pnts);
    for (int i = F.lane; i < w * h; i += 64) {
      const int ry = i / w, rx = i - ry * w;
      uint8_t* p = F.fb + (SF_TXT_BOX_Y0 + ry) * SF_IMG_W + SF_TXT_BOX_X0 + rx;
      *p = (uint8_t)sfr::text_pixel(SF_TXT_BOX_X0 + rx, SF_TXT_BOX_Y0 + ry, masks, *p);
    }
  }
  __builtin_amdgcn_wave_barrier();
  F.resample(Box{SF_TXT_BOX_X0, SF_TXT_BOX_Y0, SF_TXT_BOX_X1, SF_TXT_BOX_Y1});
}
// ---- vulnerability bar (drawVlner, :205-225): two filled rectangles, a lane per pixel of the box.  `state`: 0..10
// tenths in grey .66, 11 = full and white (kill-ready, :268)
// (the state: sfd::bar_state, sf_drawrec.h)
template <bool RESIZE>
__device__ __forceinline__ void draw_bar(const Frame<RESIZE>& F, int state) {
  const int v = state > 10 ? 10 : state, vg = state > 10 ? 255 : 168;
  constexpr int w = SF_BAR_BOX_X1 - SF_BAR_BOX_X0, h = SF_BAR_BOX_Y1 - SF_BAR_BOX_Y0;
  static_assert(SF_BAR_BOX_X0 == (int)((255 - SF_VP_X) * SF_SCALE) && SF_BAR_BOX_X1 == (int)((455 - SF_VP_X) * SF_SCALE) &&
                SF_BAR_BOX_Y0 <= (522 - SF_VP_Y) * SF_SCALE && SF_BAR_BOX_Y1 >= (532 - SF_VP_Y) * SF_SCALE, "bar box");
  for (int i = F.lane; i < w * h; i += 64) {
    const int ry = i / w, rx = i - ry * w;
    uint8_t* p = F.fb + (SF_BAR_BOX_Y0 + ry) * SF_IMG_W + SF_BAR_BOX_X0 + rx;
    *p = (uint8_t)sfr::bar_pixel(SF_BAR_BOX_X0 + rx, SF_BAR_BOX_Y0 + ry, v, vg, *p);
  }
  __builtin_amdgcn_wave_barrier();
  F.resample(Box{SF_BAR_BOX_X0, SF_BAR_BOX_Y0, SF_BAR_BOX_X1, SF_BAR_BOX_Y1});
}

// The score and the bar are drawn last, over pixels nothing else has touched (else: in place), and what they draw is
// a function of the points / of the bar's state alone: pictures drawn once per batch by the code above
// (sf_hud_kernel), like the fortress's.  One picture = the box of the surface, then the box of the 84x84 image that
// reads it, rows `row` bytes apart; at most 64 words each: a lane per word, all loads before the first store.
struct HudWords {  // a picture's two words of this lane, asked for in the prologue (hud_fetch)
  uint32_t wf, wo;
};
template <bool RESIZE>
__device__ __forceinline__ HudWords hud_fetch(const unsigned char* pic, int row, const Box b, int lane) {
  const Box o = out_box(b);
  const int wpr = row / 4, bh = b.y1 - b.y0, oh = o.y1 - o.y0, r = lane / wpr;
  const uint32_t* gf = reinterpret_cast<const uint32_t*>(pic) + lane;
  const uint32_t* go = reinterpret_cast<const uint32_t*>(pic + ((bh * row + 15) & ~15)) + lane;
  return HudWords{r < bh ? *gf : 0u, (RESIZE && r < oh) ? *go : 0u};
}
template <bool RESIZE>
__device__ __forceinline__ void hud_picture(const Frame<RESIZE>& F, unsigned char* pic, int row, const Box b, bool store,
                                            const bool with_out = true, const bool have_pre = false, const HudWords pre = HudWords{0u, 0u}) {
  const Box o = out_box(b);
  const int lane = F.lane, wpr = row / 4;
  const int bw = b.x1 - b.x0, bh = b.y1 - b.y0, ow = o.x1 - o.x0, oh = o.y1 - o.y0;
  const int r = lane / wpr, c4 = (lane - r * wpr) * 4;
  uint8_t* pf = F.fb + (b.y0 + r) * SF_IMG_W + b.x0 + c4;
  uint8_t* po = F.obuf + (o.y0 + r) * SF_OUT + o.x0 + c4;
  uint32_t* gf = reinterpret_cast<uint32_t*>(pic) + lane;
  uint32_t* go = reinterpret_cast<uint32_t*>(pic + ((bh * row + 15) & ~15)) + lane;
  if (store) {
    uint32_t wf = 0, wo = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (r < bh && c4 + k < bw) wf |= (uint32_t)pf[k] << (8 * k);
      if (RESIZE && r < oh && c4 + k < ow) wo |= (uint32_t)po[k] << (8 * k);
    }
    if (r < bh) *gf = wf;
    if (RESIZE && r < oh) *go = wo;
    return;
  }
  uint32_t wf = pre.wf, wo = pre.wo;
  if (!have_pre) {  // uniform
    wf = r < bh ? *gf : 0u;
    wo = (RESIZE && with_out && r < oh) ? *go : 0u;
  }
  // (rows of 26 .. 40 bytes, a word per lane: lane 0 holds a row's first word)
  put_row_word(pf, wf, lane_below(wf), r < bh ? bw - c4 : 0);
  if (RESIZE && with_out) put_row_word(po, wo, lane_below(wo), r < oh ? ow - c4 : 0);
  __builtin_amdgcn_wave_barrier();
  // (something within that reach, not on the box: the surface part of the picture still holds, the 84x84 pixels are taken
  //  from the surface as it is now)
  if (RESIZE && !with_out) F.resample(b);
}
__device__ __forceinline__ unsigned char* hud_score_picture(const unsigned char* hud, int pnts) {
  return const_cast<unsigned char*>(hud) + (size_t)(pnts + SF_HUD_SCORE_HALF) * SF_HUD_SCORE_BYTES;
}
__device__ __forceinline__ unsigned char* hud_bar_picture(const unsigned char* hud, int state) {
  return const_cast<unsigned char*>(hud) + (size_t)2 * SF_HUD_SCORE_HALF * SF_HUD_SCORE_BYTES + (size_t)state * SF_HUD_BAR_BYTES;
}
// (out_box of the two boxes)
constexpr int kHudScoreOutW = (SF_TXT_BOX_X1 * 14 + 14) / 15 - (SF_TXT_BOX_X0 * 14) / 15,
              kHudScoreOutH = (SF_TXT_BOX_Y1 * 21 + 22) / 23 - (SF_TXT_BOX_Y0 * 21) / 23,
              kHudBarOutW = (SF_BAR_BOX_X1 * 14 + 14) / 15 - (SF_BAR_BOX_X0 * 14) / 15,
              kHudBarOutH = SF_OUT - (SF_BAR_BOX_Y0 * 21) / 23;
static_assert(((SF_TXT_BOX_Y1 - SF_TXT_BOX_Y0) * SF_HUD_SCORE_ROW + 15) / 16 * 16 + kHudScoreOutH * SF_HUD_SCORE_ROW <= SF_HUD_SCORE_BYTES &&
              SF_HUD_SCORE_ROW >= kHudScoreOutW && SF_HUD_SCORE_ROW >= SF_TXT_BOX_X1 - SF_TXT_BOX_X0 &&
              kHudScoreOutH * (SF_HUD_SCORE_ROW / 4) <= 64 && (SF_TXT_BOX_Y1 - SF_TXT_BOX_Y0) * (SF_HUD_SCORE_ROW / 4) <= 64,
              "score picture layout");
static_assert(((SF_BAR_BOX_Y1 - SF_BAR_BOX_Y0) * SF_HUD_BAR_ROW + 15) / 16 * 16 + kHudBarOutH * SF_HUD_BAR_ROW <= SF_HUD_BAR_BYTES &&
              SF_HUD_BAR_ROW >= kHudBarOutW && SF_HUD_BAR_ROW >= SF_BAR_BOX_X1 - SF_BAR_BOX_X0 &&
              kHudBarOutH * (SF_HUD_BAR_ROW / 4) <= 64 && (SF_BAR_BOX_Y1 - SF_BAR_BOX_Y0) * (SF_HUD_BAR_ROW / 4) <= 64,
              "bar picture layout");

template <bool RESIZE>
#ifndef SF_RENDER_WPE
#define SF_RENDER_WPE 3 /* waves per SIMD the register budget is held to (168 VGPRs, some thirty spilled: measured 184 against 202 us
                            at two waves); LDS -- 14 KB per frame -- allows 11 workgroups per CU */
#endif
__global__ __launch_bounds__(64, SF_RENDER_WPE) void sf_render_kernel(SfRenderArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t fbw[kFbPadWords];
  const uint32_t* const tabw = a.tabs;  // 2.7 KB read by every wave: L1/L2 resident; LDS is better spent on waves
  __shared__ __attribute__((aligned(16))) uint32_t torw[Frame<RESIZE>::kTorWords];  // sf_tor_dev.h's records / objects / accumulators
  uint8_t* fb = reinterpret_cast<uint8_t*>(fbw);
  const int lane = threadIdx.x;
  // ---- this lane's line of the frame's draw order: lanes 0..2 the ship's three, 3..6 idle (the fortress comes from pictures
  // and alpha maps), 7..63 those of missile slots 0..18
  // ... and this lane's entry of the tap tables' period (Frame::ltab): columns 0 .. 27, then rows 0 .. 20
  uint4 lt_e = {0u, 0u, 0u, 0u};
  if (RESIZE) {
    const int ti = lane < Frame<RESIZE>::kLtabCols ? lane : min(SF_OUT + lane - Frame<RESIZE>::kLtabCols, SF_OUT + 20);
    lt_e = reinterpret_cast<const uint4*>(a.tabs)[ti];
  }
#ifndef SF_RENDER_STAGGER
#define SF_RENDER_STAGGER 0 /* A/B: the first generation's workgroups (16 per CU, all launched at once) start in four phases
                               SF_RENDER_STAGGER x 64 cycles apart, by SF_RENDER_STAGGER_SHIFT bits of the workgroup's index */
#endif
#ifndef SF_RENDER_STAGGER_SHIFT
#define SF_RENDER_STAGGER_SHIFT 10
#endif
#if SF_RENDER_STAGGER
  if (blockIdx.x < 4096u) {
    const unsigned ph = (blockIdx.x >> SF_RENDER_STAGGER_SHIFT) & 3u;  // uniform
    for (unsigned k = 0; k < ph; k++) __builtin_amdgcn_s_sleep(SF_RENDER_STAGGER);
  }
#endif
  // ---- which env (pick_env).  Nearly every workgroup behind the front draws env = its index - n_front, and learns that
  // from one word of the hint: the record's loads go out for that env at once, next to the word's load, instead of behind it.
  int env = blockIdx.x;
  bool recheck = false;
  if (a.hint) {
    if ((int)blockIdx.x >= a.n_front) {
      env = (int)blockIdx.x - a.n_front;
      recheck = true;
    } else {
      env = pick_env(a, (int)blockIdx.x, lane);
      if (env < 0) return;  // uniform, before any barrier
    }
  }
  env = __builtin_amdgcn_readfirstlane(env);
  uint8_t* const frame_out = a.out + (size_t)env * a.out_stride;

  // ---- ROUND TRIP 1: the env's draw record (sf_drawrec.h), left by the step kernel: 32 bytes of finished decisions through
  // ONE scalar load -- uniform by construction, in SGPRs without a v_readfirstlane, and every test of them below is a scalar
  // bit test -- and, per lane, the transform (x, y, cos, sin) of the object its stroke belongs to.  Round 3 read five chunks
  // of the state and three rows of the tile's missile pool here, every one of a tile's 64 frames filtering the same rows for
  // its own entries, and decided in all 64 lanes what one lane of the step kernel decides now.
  const unsigned char* const rec = a.draw + (size_t)(env >> 6) * SF_DR_TILE_BYTES + (size_t)(env & 63) * SF_DR_LANE_STRIDE;
  typedef unsigned u4s_t __attribute__((ext_vector_type(4)));
  const u4s_t hd0 = *reinterpret_cast<const __attribute__((address_space(4))) u4s_t*>(
      reinterpret_cast<const __attribute__((address_space(4))) void*>((unsigned long long)rec));
  const u4s_t hd1 = *reinterpret_cast<const __attribute__((address_space(4))) u4s_t*>(
      reinterpret_cast<const __attribute__((address_space(4))) void*>((unsigned long long)(rec + SF_DR_PIECE_STRIDE)));
  const unsigned hd[8] = {hd0.x, hd0.y, hd0.z, hd0.w, hd1.x, hd1.y, hd1.z, hd1.w};
  constexpr int kFirstMissileLane = 7;
  static_assert(kFirstMissileLane + 3 * 19 == 64, "slots 0 .. 18 fill the wave behind the ship's and the fortress's strokes");
  const int mslot = ((lane - kFirstMissileLane) * 171) >> 9;  // (lane - 7) / 3 for lanes 7 .. 63
  const int obj = lane < 3 ? SF_DR_OBJ_SHIP : (lane < kFirstMissileLane ? SF_DR_OBJ_FORT : SF_DR_OBJ_MISSILE0 + mslot);
  const d2_t tf = *reinterpret_cast<const d2_t*>(rec + (unsigned)((SF_DR_PIECE_OBJ0 + obj) * SF_DR_PIECE_STRIDE));  // position, float64
  const int mheading = *reinterpret_cast<const int16_t*>(rec + (unsigned)(SF_DR_ANGLES_OFF + 2 * max(mslot, 0)));  // (a missile's)
  const d2_t shipd = *reinterpret_cast<const d2_t*>(rec + (unsigned)((SF_DR_PIECE_OBJ0 + SF_DR_OBJ_SHIP) * SF_DR_PIECE_STRIDE));  // (uniform)
  // (a frame stack's done flag of this env: asked for here, read where the older slots are handled)
  // (through an index the compiler cannot see is the same in every lane: a byte it knows to be uniform it moves to a scalar
  //  register at once -- v_readfirstlane behind a wait, i.e. a memory round trip in front of everything else, 7 % of the
  //  step with a frame stack)
  unsigned fin_v = 0;
  if (RESIZE && a.stack_done) {
    int idx = env;
    asm volatile("" : "+v"(idx));
    fin_v = a.stack_done[idx];
  }
  if (recheck) {
    const unsigned long long w = a.hint[env >> 6];  // uniform: a scalar load, in flight beside the loads above
    if ((w >> (env & 63)) & 1ull)                    // a hinted env: one of the front workgroups may be drawing it
      if (pick_env(a, (int)blockIdx.x, lane) < 0) return;
  }
  const float ship_x = __uint_as_float(hd[SF_DRW_SHIP_X]), ship_y = __uint_as_float(hd[SF_DRW_SHIP_Y]);
  const int pnts = (int)hd[SF_DRW_POINTS];
  const unsigned objmask = (SF_RENDER_SKIP & 2) ? (hd[SF_DRW_OBJMASK] & 3u) : hd[SF_DRW_OBJMASK];
  const unsigned mmask = objmask >> SF_DR_OBJ_MISSILE0;
  const unsigned shw = hd[SF_DRW_SHELLS], smask = (SF_RENDER_SKIP & 2) ? 0u : (shw & SF_MASK_LOW);
  const unsigned fl = hd[SF_DRW_FLAGS];
  const int bgi = (int)SF_DRF_BG(fl), bstate = (int)SF_DRF_BAR(fl);
  const bool dead_ship = !(fl & SF_DRF_SHIP_ALIVE);

  // ---- ROUND TRIP 2: everything else this frame reads from memory, asked for at once now that the record says what that
  // is -- a wave's loads come back in the order they were issued, so what is needed first goes first: (1) the shells'
  // positions, for the frames that have one; (2) what is restored right after the surface is in: the dead ship's cached
  // explosion -- key and pixels together, the pixels used if the key matches; (3) the 84x84 background's seven pieces;
  // (4) LAST, the surface's direct-to-LDS loads.
  // shells (14 % of the frames have one): lane 4 s + k draws stroke k of slot s (slots 0 .. 15; the last four slots --
  // seventeen live shells -- have a late round of their own) -- unless the top lanes of the missiles' range are free for
  // them (the record's SF_DRF_MERGE_SHELLS: nearly always): then their strokes sit there, behind the missiles' as in the draw
  // order, and go through draw_strokes with everything else instead of a call, a chunk, a resample pass of their own
  const bool merge_shells = SF_MERGE_SHELLS && (fl & SF_DRF_MERGE_SHELLS);
  const int shl = merge_shells ? lane - (int)(shw >> 24) : lane;  // 4 * slot + stroke, negative = not a shell's lane
  const unsigned char* const tile = a.state + (long)(env >> 6) * sfl::kTileBytes;  // (the shells' positions live in the state)
  const int o16 = (env & 63) * 16;
  d2_t shell_p = {0.0, 0.0}, shell_v = {0.0, 0.0};
  if (smask) {  // uniform
    shell_p = R_LD(d2_t, R_CHUNK(shell_pos, max(shl, 0) >> 2), o16);
    shell_v = R_LD(d2_t, R_CHUNK(shell_vel, max(shl, 0) >> 2), o16);
  }
  unsigned char* const xc_mine = a.xcache ? a.xcache + (size_t)env * SF_XC_BYTES : nullptr;
  // ---- The frame starts as a copy of the static background: the 92x90 surface into LDS, its 84x84 image into the
  // caller's frame.  Which one -- score / bar baked in, the live fortress's picture baked in -- the record says.
  auto start_surface = [&](int variant) {
    // Ten loads that write LDS directly (global_load_lds: lane i's 16 bytes land at M0 + offset + 16 i; the offset
    // moves both addresses), all in flight at once and without registers: 8 x 1 KiB, the 5 whole pieces behind them,
    // and the last 26 words one per lane (the spare row's taps have weight zero: any byte will do, so the source
    // index is clamped into the variant).  Written as one asm statement because the compiler, once it knows of such a
    // load in flight, drains everything (vmcnt(0)) before each later store or use of a loaded value -- and left as a
    // loop through registers it waits for each 16 bytes before it asks for the next, nine round trips one after the
    // other.  Unknown to the compiler they only make its own counted waits stricter; the wait that matters, before the
    // surface is first read, is the explicit one in front of the barrier below.
    static_assert(kFbVec == 8 * 64 + 5 && kFbPadWords - 4 * kFbVec == 26 && SF_BG_STRIDE / 4 > 4 * kFbVec,
                  "the copy below is written out for the 92x90 surface");
    const char* bgv = reinterpret_cast<const char*>(a.bg) + variant * SF_BG_STRIDE;
    const char* p0 = bgv + lane * 16;
    const char* p1 = p0 + 4096;
    const char* p2 = p0 + 8192;
    const char* p3 = bgv + 4 * min(4 * kFbVec + lane, SF_BG_STRIDE / 4 - 1) - (4 * 4 * kFbVec - 8192);
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)fbw;
    unsigned keep_m0;
    unsigned long long keep_exec;
    asm volatile(
        "s_mov_b32 %[km], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[p0], off\n\t"
        "global_load_lds_dwordx4 %[p0], off offset:1024\n\t"
        "global_load_lds_dwordx4 %[p0], off offset:2048\n\t"
        "global_load_lds_dwordx4 %[p0], off offset:3072\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[p1], off\n\t"
        "global_load_lds_dwordx4 %[p1], off offset:1024\n\t"
        "global_load_lds_dwordx4 %[p1], off offset:2048\n\t"
        "global_load_lds_dwordx4 %[p1], off offset:3072\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_mov_b64 %[ke], exec\n\t"
        "s_and_b64 exec, %[ke], 0x1f\n\t"
        "global_load_lds_dwordx4 %[p2], off\n\t"
        "s_and_b64 exec, %[ke], 0x3ffffff\n\t"
        "global_load_lds_dword %[p3], off offset:80\n\t"
        "s_mov_b64 exec, %[ke]\n\t"
        "s_mov_b32 m0, %[km]"
        : [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec)
        : [lds] "s"(lds), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3)
        : "memory", "scc");
  };
  // the score's and the bar's pictures, when they will not be the baked-in 0000000 / empty ones: used last, asked for now
  const Box tbox{SF_TXT_BOX_X0, SF_TXT_BOX_Y0, SF_TXT_BOX_X1, SF_TXT_BOX_Y1};
  const Box bbox{SF_BAR_BOX_X0, SF_BAR_BOX_Y0, SF_BAR_BOX_X1, SF_BAR_BOX_Y1};
#ifndef SF_HUD_PREFETCH
#define SF_HUD_PREFETCH 1
#endif
  const bool score_pre = SF_HUD_PREFETCH && a.hud && pnts != 0 && pnts >= -SF_HUD_SCORE_HALF && pnts < SF_HUD_SCORE_HALF && !(SF_RENDER_SKIP & 4);
  const bool bar_pre = SF_HUD_PREFETCH && a.hud && bstate != 0 && !(SF_RENDER_SKIP & 8);
  // (in front of the background's pieces, for what is restored right behind the barrier or at the very end: the dead ship's
  //  cached explosion -- key and pixels together, the pixels used if the key matches --, the score's and the bar's pictures)
  XcFetch xf = {};
  if (dead_ship && xc_mine) xf = xc_fetch<RESIZE>(xc_mine, lane);
  HudWords hscore = {}, hbar = {};
  if (score_pre) hscore = hud_fetch<RESIZE>(hud_score_picture(a.hud, pnts), SF_HUD_SCORE_ROW, tbox, lane);
  if (bar_pre) hbar = hud_fetch<RESIZE>(hud_bar_picture(a.hud, bstate), SF_HUD_BAR_ROW, bbox, lane);
  // the 84x84 background's seven pieces
  Pieces frame0 = {};
  if (RESIZE && !(SF_RENDER_SKIP & 8192)) frame0 = load_pieces(reinterpret_cast<const uint4*>(a.bg84 + bgi * (kOutBytes / 4)), kOutBytes / 16, lane);  // (bit 13: timing-only, no background copy)
  // ... and LAST of the round trip, the surface's ten direct-to-LDS loads: a wave's loads come back in issue order and the
  // compiler does not know of these ten, so every wait it counts out for something issued before them stays a wait for
  // that alone.
  if (!(SF_RENDER_SKIP & 4096)) start_surface(bgi);  // (bit 12: timing-only, no surface)

  // ---- the shells' strokes, built here with the surface's loads in flight
  sft::Quad sq0 = {};
  bool sq0_valid = false;
  auto shell_quad = [&](d2_t s, d2_t v, int k, bool have, sft::Quad* q) -> bool {
    const double dx = s.x - sfc::fort_x, dy = s.y - sfc::fort_y;
    const bool valid = have && sqrt(dx * dx + dy * dy) > 21.0;  // drawn only once clear of the fortress (SRC/draw.cpp:249-250)
    int ideg = 0;
    if (valid) {
      // mAngle = stdAngle(rad2deg(atan2(dy, dx))) at launch (SRC/game.cpp:263); the velocity kept in
      // the state has that direction.  drawWireFrame takes it as an int (truncation).
      double ang = atan2(v.y, v.x) * 180.0 / M_PI;
      if (ang < 0) ang += 360.0;
      ideg = (int)ang;
    }
    if (valid) *q = line_quad(shell_seg(k), s.x, s.y, a.trig, ideg);
    return valid;
  };
  if (smask) {  // uniform
    asm volatile("" : "+v"(shell_p.x), "+v"(shell_p.y), "+v"(shell_v.x), "+v"(shell_v.y));
    sq0_valid = shell_quad(shell_p, shell_v, lane & 3, shl >= 0 && ((smask >> (shl >> 2)) & 1u), &sq0);
  }
  // the 84x84 background's seven stores, behind everything: stores count like loads, in the same order -- in front of the
  // surface's loads, the wait for the surface would be a wait for their acknowledgement from HBM as well
  if (RESIZE && !(SF_RENDER_SKIP & 8192)) {
    if (SF_FRAME_NT == 2 || (SF_FRAME_NT == 1 && a.out_stride > (size_t)kOutBytes))  // uniform: a slot of a frame stack (SF_FRAME_NT)
      store_pieces<true>(frame0, reinterpret_cast<uint4*>(frame_out), kOutBytes / 16, lane);
    else
      store_pieces<false>(frame0, reinterpret_cast<uint4*>(frame_out), kOutBytes / 16, lane);
  }
  // the frame stack's older slots
  // (the done flag is looked at where it costs nothing: behind the wait for the surface, below -- asked for in front of it, the
  //  compiler's wait for that one byte counts only the loads it knows of and ends up waiting for part of the surface's ten as
  //  well, with the strokes' arithmetic still to come: 1.4 us per step of a frame stack)
  const bool stack_traffic = RESIZE && a.stack_prev;  // more loads / stores behind the seven: see the wait below
  if (RESIZE && a.stack_prev) {
    const bool fin = a.stack_done && __builtin_amdgcn_readfirstlane((int)fin_v) != 0;
    const uint4 z = {0u, 0u, 0u, 0u};
    const uint4* src = reinterpret_cast<const uint4*>(a.stack_prev + (size_t)env * a.out_stride + kOutBytes);
    uint4* dst = reinterpret_cast<uint4*>(frame_out - (ptrdiff_t)a.stack_slot * kOutBytes);
    const int n = (a.stack_n - 1) * (kOutBytes / 16);
    if (fin) {
      for (int i = lane; i < n; i += 64) dst[i] = z;
    } else {
      // a frame's worth of loads in flight, then its stores (a plain loop waits for each 16 bytes before the next load)
      for (int base = 0; base < n; base += 64 * kFrameRounds) copy_pieces(src + base, dst + base, n - base, lane);
    }
  }

  // ---- the frame's strokes in draw order, one per lane: the live ship's three, the fortress's four when it has to be
  // drawn in place, the missiles' (SRC/draw.cpp:233-247) -- the lane's corners under the object's transform, translate(pos)
  // rotate(angle) (drawWireFrame, SRC/draw.cpp:112-129), in device pixels
  const bool fort_lane = lane >= 3 && lane < kFirstMissileLane;
  const bool svalid = ((objmask >> obj) & 1u) && !fort_lane && !((SF_RENDER_SKIP & 1) && lane < kFirstMissileLane);
  const int sobj = lane < 3 ? 0 : (lane < kFirstMissileLane ? 3 : kFirstMissileLane + 3 * mslot);
  sft::Quad mq = {};
  if (svalid) {
    const int k = lane < 3 ? lane : lane - kFirstMissileLane - 3 * mslot;
    const int heading = lane < 3 ? (int)(int16_t)(hd[SF_DRW_ANGLES] & 0xFFFFu) : mheading;
    mq = line_quad(lane < 3 ? ship_seg(k) : missile_seg(k), tf.x, tf.y, a.trig, heading);
  }
  bool dvalid = svalid;  // what draw_strokes is given: with the shells' strokes in the top lanes when they fit there
  int dobj = sobj, dkind = sftd::kKindLines3;
  if (merge_shells && shl >= 0) {
    mq = sq0;
    dvalid = sq0_valid;
    dobj = lane & ~3;
    dkind = sftd::kKindShell;
  }
  // (a shell is drawn or not as a whole: its four lanes agree on `valid`)

  // The 84x84 background's seven stores are the LAST vector-memory instructions in front of this wait (but for a frame
  // stack's shifted / cleared slots): vmcnt counts loads, stores and LDS-DMA together in issue order, so `vmcnt(7)` =
  // everything but these stores is done -- the surface is in LDS -- without waiting for the stores to be acknowledged.  The
  // byte stores that follow land on top of them anyway: one wave's stores to one address are performed in program order.
  static_assert(kFrameRounds == 7 && 6 * 64 < kOutBytes / 16, "all seven stores have lanes to do");
  if (RESIZE && !stack_traffic && !(SF_RENDER_SKIP & 8192)) {
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // a frame stack's finished env (`current_obs *= masks`, rl/train.py:92-93): its other slots are zeroed
  if (RESIZE && a.stack_done && !a.stack_prev) {
    if (__builtin_amdgcn_readfirstlane((int)fin_v) != 0) {
      const uint4 z = {0u, 0u, 0u, 0u};
      for (int sl = 0; sl < a.stack_n; sl++) {
        if (sl == a.stack_slot) continue;
        uint4* dst = reinterpret_cast<uint4*>(frame_out + (ptrdiff_t)(sl - a.stack_slot) * kOutBytes);
        for (int i = lane; i < kOutBytes / 16; i += 64) dst[i] = z;
      }
    }
  }
  const Frame<RESIZE> F{fb, frame_out, tabw, nullptr, lane, torw, torw + Frame<RESIZE>::kLtabAt, lt_e};
  if (SF_RENDER_STOP == 1) return;

  // (sf_tor_dev.h's accumulators start out zero and every call leaves them so; what lies below them -- records, headers,
  //  objects, task list, map -- is written by every call before it is read: twenty rounds of stores a frame that bought nothing)
#ifndef SF_TOR_ZERO_ALL
#define SF_TOR_ZERO_ALL 0
#endif
  for (int i = (SF_TOR_ZERO_ALL ? 0 : sftd::kAccAtF) + lane; i < Frame<RESIZE>::kTorWords; i += 64) torw[i] = 0u;
  __builtin_amdgcn_wave_barrier();
  // ---- what is restored from round trip 2's registers: the dead ship's explosion if its cache entry is this one
  // (the entry is keyed by where the ship died: the float64 position the picture is a function of)
  const double ship_xd = shipd.x, ship_yd = shipd.y;
  XcState xst{0u, 0, 0};
  bool explosion_done = false;
  const bool explosion = dead_ship && !(SF_RENDER_SKIP & (1 | 256));
  if (explosion && xc_mine) explosion_done = xc_apply(F, xf, ship_xd, ship_yd, &xst);
  if (SF_RENDER_STOP == 2) return;
  // ---- ship (SRC/draw.cpp:233-237): a dead ship's explosion that was not in the cache is the first thing drawn
  if (explosion && !explosion_done) xst = ship_explosion(F, a.arcs, xc_mine, ship_xd, ship_yd, true, /*skip_lookup=*/true);
  // ---- fortress (:238-242), destroyed: it explodes for 1000 ms where it stands: one more picture drawn once per batch, in
  // the layout of the per-env explosion cache (a trained agent destroys it every few seconds -- 30 frames each time).
  // Restored when what the ship drew stays clear of it (wider by the reach: what its 84x84 pixels read) -- the two touch no
  // pixel in common then, so it may go in before the ship --; else drawn in place between the ship and the missiles.
  if ((fl & SF_DRF_FORT_EX_PATCH) && !(SF_RENDER_SKIP & (1 | 512)))
    ship_explosion(F, a.arcs, const_cast<unsigned char*>(a.fpatch) + 36 * SF_FP_BYTES, sfc::fort_x, sfc::fort_y, false);
  const bool fort_explodes_in_place = (fl & SF_DRF_FORT_EX_PLACE) && !(SF_RENDER_SKIP & (1 | 512));
  if (SF_RENDER_STOP == 3) return;
  // ---- the live ship, the fortress in place, the missiles (:233-247): all their strokes at once
  const bool fort_lines_in_place = (objmask >> SF_DR_OBJ_FORT) & 1u;
  if (!fort_explodes_in_place && !fort_lines_in_place) {
    F.draw_strokes(mq, dvalid, dobj, dkind);
    if (SF_RENDER_STOP > 40) return;
  } else {  // (rare: the ship, or its explosion, next to the fortress or its explosion: the reference's order, one by one)
    F.draw_strokes(mq, dvalid && lane < 3, dobj, dkind);
    if (fort_explodes_in_place) draw_explosion(F, a.arcs, sfc::fort_x, sfc::fort_y);
    else fort_in_place(F, a.falpha, min(max((int)(int16_t)(hd[SF_DRW_ANGLES] >> 16) / 10, 0), 35));
    F.draw_strokes(mq, dvalid && lane >= kFirstMissileLane, dobj, dkind);
  }
  if ((fl & SF_DRF_MISSILE19) && !(SF_RENDER_SKIP & 2)) {  // (the twentieth missile: its strokes have no lanes of their own)
    const d2_t t19 = *reinterpret_cast<const d2_t*>(rec + (SF_DR_PIECE_OBJ0 + SF_DR_OBJ_MISSILE0 + 19) * SF_DR_PIECE_STRIDE);
    const int h19 = *reinterpret_cast<const int16_t*>(rec + SF_DR_ANGLES_OFF + 2 * 19);
    const sft::Quad q19 = line_quad(missile_seg(lane < 3 ? lane : 0), t19.x, t19.y, a.trig, h19);
    F.draw_strokes(q19, lane < 3, 0, sftd::kKindLines3);
  }
  if (SF_RENDER_STOP == 4) return;
  // ---- shells (:248-253): slot order
  if (smask && !merge_shells) {
    F.draw_strokes(sq0, sq0_valid, lane & ~3, sftd::kKindShell);
    if (smask >> 16) {  // (slots 16 .. 19)
      const int slot = 16 + (lane >> 2);
      const bool have = lane < 16 && ((smask >> slot) & 1u);
      sft::Quad sq = {};
      const d2_t s = R_LD(d2_t, R_CHUNK(shell_pos, have ? slot : 0), o16);
      const d2_t v = R_LD(d2_t, R_CHUNK(shell_vel, have ? slot : 0), o16);
      const bool valid = shell_quad(s, v, lane & 3, have, &sq);
      F.draw_strokes(sq, valid, lane & ~3, sftd::kKindShell);
    }
  }
  if (SF_RENDER_STOP == 5) return;
  // ---- score and bar, last (SRC/draw.cpp:266-268): baked into the background already (0000000 / empty), or one of
  // the pictures, or -- something else touches their pixels, or the points are off the table -- in place.
  // A score / bar that is not baked in is one of the pictures (hud_picture) unless something comes within reach (kReachX,
  // kReachY) of its box: the picture's 84x84 pixels read that far, and it is restored after everything else was resampled.
  // When the only thing on them is the dead ship's explosion (a ship lost through the upper or lower edge of the big
  // hexagon, right under the score / over the bar: a sixth of the losses each, 30 frames a time), what the box ends up as
  // belongs to (where the ship died, the points / the bar's state): drawn in place once, kept in the env's explosion
  // cache entry, restored for the other frames.
  // (one call site each for drawing and for the picture copy: what to do is decided first)
  if (!(fl & SF_DRF_BAKED_TEXT) && !(SF_RENDER_SKIP & 4)) {
    constexpr unsigned kBits = RESIZE ? 12u : 4u;
    const bool near_text = fl & SF_DRF_NEAR_TEXT, ex_text = fl & SF_DRF_EX_TEXT, other_text = fl & SF_DRF_OTHER_TEXT;
    unsigned char* pic = nullptr;
    bool draw = true, save = false, with_out = true;
    bool pre = false;
    if (a.hud && !near_text && pnts >= -SF_HUD_SCORE_HALF && pnts < SF_HUD_SCORE_HALF) {
      pic = hud_score_picture(a.hud, pnts);
      draw = false;
      with_out = !(ex_text || other_text);
      pre = score_pre;
    } else if (xst.flags && ex_text && !other_text) {
      pic = xc_mine + kXcScore;
      draw = save = !((xst.flags & kBits) == kBits && xst.points == pnts);
    }
    if (draw) draw_score(F, pnts, a.glyphs);
    if (pic) hud_picture(F, pic, SF_HUD_SCORE_ROW, tbox, save, with_out, pre, hscore);
    if (save) {
      xst.flags |= kBits;
      if (lane == 0) {
        *reinterpret_cast<int*>(xc_mine + kXcHudKeys) = pnts;
        *reinterpret_cast<unsigned*>(xc_mine + kXcFlags) = xst.flags;
      }
    }
  }
  if (!(fl & SF_DRF_BAKED_BAR) && !(SF_RENDER_SKIP & 8)) {
    constexpr unsigned kBits = RESIZE ? 48u : 16u;
    const bool near_bar = fl & SF_DRF_NEAR_BAR, ex_bar = fl & SF_DRF_EX_BAR, other_bar = fl & SF_DRF_OTHER_BAR;
    const int state = bstate;
    unsigned char* pic = nullptr;
    bool draw = true, save = false, with_out = true, pre = false;
    if (a.hud && !near_bar) {
      pic = hud_bar_picture(a.hud, state);
      draw = false;
      with_out = !(ex_bar || other_bar);
      pre = bar_pre;
    } else if (xst.flags && ex_bar && !other_bar) {
      pic = xc_mine + kXcBar;
      draw = save = !((xst.flags & kBits) == kBits && xst.bar == state);
    }
    if (draw) draw_bar(F, state);
    if (pic) hud_picture(F, pic, SF_HUD_BAR_ROW, bbox, save, with_out, pre, hbar);
    if (save) {
      xst.flags |= kBits;
      if (lane == 0) {
        *reinterpret_cast<int*>(xc_mine + kXcHudKeys + 4) = state;
        *reinterpret_cast<unsigned*>(xc_mine + kXcFlags) = xst.flags;
      }
    }
  }
  __syncthreads();

  // ---- epilogue: the 84x84 frame is already where it belongs; the raw one leaves LDS (8280 = 1035 * 8)
  if (!RESIZE) {
    const uint2* src = reinterpret_cast<const uint2*>(fbw);
    uint2* out = reinterpret_cast<uint2*>(frame_out);
    for (int i = lane; i < kFbBytes / 8; i += 64) out[i] = src[i];
  }
}

// one workgroup per sector: the live fortress on the bare background -- its alpha map (the host's: sf_image_fort_alpha)
// lerped in like a frame does in place --, its box of the surface and of the 84x84 image saved for fort_patch_copy; workgroup 36:
// the destroyed fortress's explosion, drawn by the frames' own code
__global__ __launch_bounds__(64) void sf_fort_patch_kernel(const uint32_t* bg, const uint32_t* bg84, const uint32_t* tabs,
                                                           unsigned char* fpatch, const double* arcs, const unsigned char* falpha) {
  __shared__ __attribute__((aligned(16))) uint32_t fbw[kFbPadWords];
  __shared__ __attribute__((aligned(16))) uint32_t obufw[kOutBytes / 4];
  __shared__ __attribute__((aligned(16))) uint32_t tabw[SF_TAB_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t torw[Frame<true>::kTorWords];
  const int lane = threadIdx.x, sector = blockIdx.x;
  for (int i = lane; i < kFbPadWords; i += 64) fbw[i] = i < kFbWords ? bg[i] : 0u;
  for (int i = lane; i < kOutBytes / 4; i += 64) obufw[i] = bg84[i];
  for (int i = lane; i < SF_TAB_WORDS; i += 64) tabw[i] = tabs[i];
  for (int i = lane; i < Frame<true>::kTorWords; i += 64) torw[i] = 0u;
  __syncthreads();
  const Frame<true> F{reinterpret_cast<uint8_t*>(fbw), reinterpret_cast<uint8_t*>(obufw), tabw, nullptr, lane, torw, nullptr, uint4{0u, 0u, 0u, 0u}};
  if (sector == 36) {  // the destroyed fortress's explosion, behind the 36 headings
    ship_explosion(F, arcs, fpatch + 36 * SF_FP_BYTES, sfc::fort_x, sfc::fort_y);  // (a zeroed entry: draws and fills it)
    return;
  }
  fort_in_place(F, falpha, sector);
  // (the rest of the patch keeps the background's values, exactly as when the fortress is drawn in place)
  __syncthreads();
  fort_patch_copy(F, fpatch + sector * SF_FP_BYTES, true);
}

// The backgrounds with the live fortress in them: 36 headings x 4 variants behind the four plain ones (index 4 (1 + sector) +
// variant, SF_BG_COUNT in all: 1.2 MB of surfaces, 1.0 MB of 84x84 images).  The fortress's picture is bytes of the surface
// and of the 84x84 image (fort_patch_copy), the score and the bar are elsewhere: a frame whose fortress picture is good starts
// from the background that already holds it -- the picture used to be three more loads in the prologue and three branchy
// partial-word writes behind the barrier, 65 instructions of four frames in five.
__global__ __launch_bounds__(256) void sf_bg_fort_kernel(uint32_t* bg, uint32_t* bg84, const unsigned char* fpatch) {
  const int sector = blockIdx.x >> 2, v = blockIdx.x & 3, t = threadIdx.x;
  const uint32_t* src = bg + v * (SF_BG_STRIDE / 4);
  uint32_t* dst = bg + (4 * (1 + sector) + v) * (SF_BG_STRIDE / 4);
  const uint32_t* src84 = bg84 + v * (kOutBytes / 4);
  uint32_t* dst84 = bg84 + (4 * (1 + sector) + v) * (kOutBytes / 4);
  for (int i = t; i < SF_BG_STRIDE / 4; i += 256) dst[i] = src[i];
  for (int i = t; i < kOutBytes / 4; i += 256) dst84[i] = src84[i];
  __syncthreads();
  const unsigned char* gp = fpatch + sector * SF_FP_BYTES;
  const Box b{kFpX0, kFpY0, kFpX1, kFpY1}, o = out_box(b);
  uint8_t* d8 = reinterpret_cast<uint8_t*>(dst);
  uint8_t* d84 = reinterpret_cast<uint8_t*>(dst84);
  if (t < 256) d8[(kFpY0 + (t >> 4)) * SF_IMG_W + kFpX0 + (t & 15)] = gp[t];
  const int ow = o.x1 - o.x0, oh = o.y1 - o.y0;
  for (int i = t; i < kFpOutRow * oh; i += 256) {
    const int r = i / kFpOutRow, c = i - r * kFpOutRow;
    if (c < ow) d84[(o.y0 + r) * SF_OUT + o.x0 + c] = gp[kFpOutAt + i];
  }
}
hipError_t sf_launch_fort_patches(uint32_t* bg, uint32_t* bg84, const uint32_t* tabs, unsigned char* fpatch, const double* arcs,
                                  const unsigned char* falpha, hipStream_t stream) {
  hipLaunchKernelGGL(sf_fort_patch_kernel, dim3(37), dim3(64), 0, stream, bg, bg84, tabs, fpatch, arcs, falpha);
  hipLaunchKernelGGL(sf_bg_fort_kernel, dim3(36 * 4), dim3(256), 0, stream, bg, bg84, fpatch);
  return hipGetLastError();
}

// one workgroup per picture: the score for blockIdx - SF_HUD_SCORE_HALF points, then the bar's states, drawn on the
// bare background by the frames' own code and saved for hud_picture
__global__ __launch_bounds__(64) void sf_hud_kernel(const uint32_t* bg, const uint32_t* bg84, const uint32_t* tabs,
                                                    unsigned char* hud, const SfGlyphAtlas* glyphs) {
  __shared__ __attribute__((aligned(16))) uint32_t fbw[kFbPadWords];
  __shared__ __attribute__((aligned(16))) uint32_t obufw[kOutBytes / 4];
  __shared__ __attribute__((aligned(16))) uint32_t tabw[SF_TAB_WORDS];
  const int lane = threadIdx.x, pic = blockIdx.x;
  for (int i = lane; i < kFbPadWords; i += 64) fbw[i] = i < kFbWords ? bg[i] : 0u;
  for (int i = lane; i < kOutBytes / 4; i += 64) obufw[i] = bg84[i];
  for (int i = lane; i < SF_TAB_WORDS; i += 64) tabw[i] = tabs[i];
  __syncthreads();
  const Frame<true> F{reinterpret_cast<uint8_t*>(fbw), reinterpret_cast<uint8_t*>(obufw), tabw, nullptr, lane, nullptr, nullptr, uint4{0u, 0u, 0u, 0u}};
  if (pic < 2 * SF_HUD_SCORE_HALF) {
    const int pnts = pic - SF_HUD_SCORE_HALF;
    draw_score(F, pnts, glyphs);
    __syncthreads();
    hud_picture(F, hud_score_picture(hud, pnts), SF_HUD_SCORE_ROW, Box{SF_TXT_BOX_X0, SF_TXT_BOX_Y0, SF_TXT_BOX_X1, SF_TXT_BOX_Y1}, true);
  } else {
    const int state = pic - 2 * SF_HUD_SCORE_HALF;
    draw_bar(F, state);
    __syncthreads();
    hud_picture(F, hud_bar_picture(hud, state), SF_HUD_BAR_ROW, Box{SF_BAR_BOX_X0, SF_BAR_BOX_Y0, SF_BAR_BOX_X1, SF_BAR_BOX_Y1}, true);
  }
}

hipError_t sf_launch_hud_pictures(const uint32_t* bg, const uint32_t* bg84, const uint32_t* tabs, unsigned char* hud,
                                  const SfGlyphAtlas* glyphs, hipStream_t stream) {
  hipLaunchKernelGGL(sf_hud_kernel, dim3(2 * SF_HUD_SCORE_HALF + SF_HUD_BAR_STATES), dim3(64), 0, stream, bg, bg84, tabs, hud, glyphs);
  return hipGetLastError();
}

// current_obs *= masks (rl/train.py:92-93) for a [n][bytes_per_env] uint8 stack: only finished envs are touched
__global__ __launch_bounds__(256) void sf_stack_clear_kernel(uint8_t* stack, size_t bytes_per_env, const uint8_t* done, int n) {
  const int env = blockIdx.x;
  if (env >= n || !done[env]) return;
  uint4* p = reinterpret_cast<uint4*>(stack + (size_t)env * bytes_per_env);
  const uint4 z = {0u, 0u, 0u, 0u};
  for (size_t i = threadIdx.x; i < bytes_per_env / 16; i += 256) p[i] = z;
}

hipError_t sf_launch_stack_clear(uint8_t* stack, size_t bytes_per_env, const uint8_t* done, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(sf_stack_clear_kernel, dim3((unsigned)n), dim3(256), 0, stream, stack, bytes_per_env, done, n);
  return hipGetLastError();
}

// ---- The explosion PRE-PASS (small batches).  The first frame of a dead ship's explosion is eight calls of the rasteriser -- seven
// rings of twelve arcs and the circle, 13 800 vector instructions -- and, drawn by the frame kernel, eight calls ONE AFTER THE OTHER
// in one wave: 130 us, latency-bound even alone on its SIMD.  A batch of 16 384 envs hides that wave behind the other frames (it
// starts first: pick_env); a batch that fits the chip in one round of waves does not: stepping took 137 us whatever the size
// (profiles/r06_render_phases.md).  Here a workgroup of EIGHT waves takes an env whose ship died in the last tick (the step
// kernel's hint words), every wave rasterises one of the eight calls into accumulators of its own, the waves composite in draw
// order onto the bare background, and the workgroup leaves the env's explosion-cache entry exactly as ship_explosion(fill) would:
// the frame kernel that follows finds the entry and copies.  What lies under an explosion's box (and within reach of it) is the
// bare background in every variant the frame kernel starts from (sf_drawrec.h: ex_text / ex_bar / fort_pic), so the entry is the
// same.  Not hinted, no cache, a box that does not fit an entry: nothing happens here and the frame kernel draws as before.
constexpr int kPreWaves = 8;
__global__ __launch_bounds__(64 * kPreWaves) void sf_explosion_kernel(SfRenderArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t fbw[kFbPadWords];
  __shared__ __attribute__((aligned(16))) uint32_t torw[kPreWaves][Frame<true>::kTorWords];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int env = pick_env(a, (int)blockIdx.x, lane);  // the blockIdx-th hinted env (every wave works it out: the same value)
  if (env < 0 || !a.xcache) return;  // (uniform over the workgroup, before any barrier)
  const unsigned char* const rec = a.draw + (size_t)(env >> 6) * SF_DR_TILE_BYTES + (size_t)(env & 63) * SF_DR_LANE_STRIDE;
  const unsigned fl = *reinterpret_cast<const unsigned*>(rec + SF_DR_PIECE_STRIDE + 4 * (SF_DRW_FLAGS - 4));
  if (fl & SF_DRF_SHIP_ALIVE) return;
  const d2_t shipd = *reinterpret_cast<const d2_t*>(rec + (unsigned)((SF_DR_PIECE_OBJ0 + SF_DR_OBJ_SHIP) * SF_DR_PIECE_STRIDE));
  const double x = shipd.x, y = shipd.y;
  unsigned char* const xc = a.xcache + (size_t)env * SF_XC_BYTES;
  {
    const double kx = *reinterpret_cast<const double*>(xc + kXcKey), ky = *reinterpret_cast<const double*>(xc + kXcKey + 8);
    const unsigned xfl = *reinterpret_cast<const unsigned*>(xc + kXcFlags);
    if (kx == x && ky == y && (xfl & 3u) == 3u) return;  // the entry is this explosion's already
  }
  const Box b = explosion_box((float)x, (float)y), o = out_box(b);
  const bool fits = b.x1 - b.x0 <= kXcRow && b.y1 - b.y0 <= kXcFbRows && o.x1 - o.x0 <= kXcRow && o.y1 - o.y0 <= kXcOutRows &&
                    b.x1 - b.x0 >= 16 && o.x1 - o.x0 >= 16;
  if (!fits) return;
  // the bare background (variant 0) into the shared surface; every wave's accumulators start out zero
  for (int i = threadIdx.x; i < kFbPadWords; i += 64 * kPreWaves) fbw[i] = i < kFbWords ? a.bg[i] : 0u;
  for (int i = sftd::kAccAtF + lane; i < Frame<true>::kTorWords; i += 64) torw[wave][i] = 0u;
  __syncthreads();
  uint8_t* const fb = reinterpret_cast<uint8_t*>(fbw);
  const sftd::CtxF C{torw[wave], fb, SF_IMG_W, SF_IMG_H, lane};
  sftd::RasterCarry carry;
  {
    const RingCall c = explosion_call(wave, lane, a.arcs, x, y);
    sftd::raster_fast<1>(C, c.q, c.valid, c.obj0, c.kind, c.grey, &carry);
  }
  for (int r = 0; r < kPreWaves; r++) {  // the reference's order: ring by ring, the circle last
    if (wave == r) sftd::raster_fast_pixels(C, carry.nobj, carry.tot_pix);
    __syncthreads();
  }
  // the entry, as ship_explosion(fill) leaves it: the box of the surface, the box of the 84x84 image that reads it, key, flags
  for (int i = threadIdx.x; i < kXcFbRows * kXcRow; i += 64 * kPreWaves) {
    const int r = i / kXcRow, c = i - r * kXcRow;
    if (r < b.y1 - b.y0 && c < b.x1 - b.x0) xc[kXcFb + i] = fb[(b.y0 + r) * SF_IMG_W + b.x0 + c];
  }
  if (wave == 0) {
    const Frame<true> F{fb, nullptr, a.tabs, nullptr, lane, torw[0], nullptr, uint4{0u, 0u, 0u, 0u}};
    F.resample_into(b, xc + kXcOut, kXcRow, o.x0, o.y0);
  }
  if (threadIdx.x == 0) {
    *reinterpret_cast<double*>(xc + kXcKey) = x;
    *reinterpret_cast<double*>(xc + kXcKey + 8) = y;
    *reinterpret_cast<unsigned*>(xc + kXcFlags) = 3u;
  }
}

hipError_t sf_launch_render(const unsigned char* state, const unsigned char* draw, int n_envs, const uint32_t* bg, const uint32_t* bg84,
                            const uint32_t* tabs, uint8_t* out, size_t out_stride, unsigned char* xcache,
                            const unsigned char* fpatch, int resize, const uint8_t* stack_done, int stack_slot, int stack_n,
                            const uint8_t* stack_prev, const unsigned long long* hint, const unsigned char* hud,
                            const double* trig, const double* arcs, const unsigned char* falpha, const SfGlyphAtlas* glyphs,
                            hipStream_t stream) {
  if (n_envs <= 0) return hipSuccess;
  // the front of the grid: a sixteenth of the batch (ships die in about 1.3 % of the ticks of random play); batches
  // whose hint words no longer fit a short scan (> 32 per lane) are drawn in env order
  const int n_front = hint && n_envs <= 64 * 64 * 32 ? (n_envs / 16 > 64 ? n_envs / 16 : 64) : 0;
  SfRenderArgs a{state, draw, n_envs, bg, bg84, tabs, out, out_stride, xcache, fpatch, stack_done, stack_slot, stack_n, stack_prev,
                 n_front ? hint : nullptr, n_front, hud, trig, arcs, falpha, glyphs};
  // the explosion pre-pass where a fresh explosion would be the launch's pole: batches that fit the chip in about two rounds of
  // waves (see sf_explosion_kernel); SFMI_EXPLOSION_PREPASS=0 / =1 in the environment: never / always (diagnostics, A/B)
  static const int prepass_env = [] { const char* e = getenv("SFMI_EXPLOSION_PREPASS"); return e ? atoi(e) : -1; }();
  if (n_front && xcache && (prepass_env > 0 || (prepass_env < 0 && n_envs <= SF_PREPASS_MAX_ENVS)))
    hipLaunchKernelGGL(sf_explosion_kernel, dim3((unsigned)n_front), dim3(64 * kPreWaves), 0, stream, a);
  const unsigned grid = (unsigned)(n_envs + n_front);
  if (resize)
    hipLaunchKernelGGL(sf_render_kernel<true>, dim3(grid), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL(sf_render_kernel<false>, dim3(grid), dim3(64), 0, stream, a);
  return hipGetLastError();
}
