// sf_raster.h -- geometry of the image observation, shared by the HIP render kernel and the host
// (static background, resize tables).  What is drawn, where, in which order and colour follows the
// reference renderer (SRC/draw.cpp:227-270, SRC/wireframe.cpp:8-70, ENV:50-58,203-206,
// rl/envs.py:28-30); HOW a stroke becomes pixel coverage is cairo's business and cairo is not in
// this image, so the anti-aliasing here is our own model -- exact area coverage of the stroke
// rectangles -- and pixel parity with cairo + cv2 is UNPINNED ("image observation").
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define SF_HD __host__ __device__ __forceinline__
#else
#define SF_HD inline
#endif

#define SF_IMG_W 90   /* int(450 * .2), ENV:57 */
#define SF_IMG_H 92   /* int(460 * .2), ENV:58 */
#define SF_OUT 84     /* rl/envs.py:29 */
#define SF_VP_X 130.0 /* viewport (130, 80, 450, 460), ENV:50 */
#define SF_VP_Y 80.0
#define SF_SCALE 0.2  /* ENV:50 */
#define SF_LINE_W 3.0 /* ls = 3 user units (ENV:50), i.e. 0.6 device pixels */

// Score text (drawScore, SRC/draw.cpp:190-203): "%07d", bold monospace 30 user units, centred on
// (355, 97), grey .5.  No font rasteriser here: seven-segment glyphs in a cell of the same metrics,
// user units: advance 18, ink 16 x 22 starting 3 into the cell, stem 5 -- calibrated so that the ink box of
// the seven characters is where the reference's own screenshot has it (rl/imgs/screens.png: 295..419 x
// 85..107; tests/test_image_host.py::test_layout_matches_the_references_own_screenshot).  At .2 scale a
// glyph is 3.2 x 4.4 pixels either way.
#define SF_TXT_ADV 18.0f
#define SF_TXT_X0 (355.0f - 3.5f * SF_TXT_ADV)
#define SF_TXT_PAD 3.0f
#define SF_TXT_W 16.0f
#define SF_TXT_H 22.0f
#define SF_TXT_T 5.0f
#define SF_TXT_TOP (97.0f - 0.5f * SF_TXT_H)

// Device copy of the INTER_AREA tables (uint32 words, four per entry):
//   [0, 4*84)       per destination column dx: {first source column, alpha0, alpha1, 0}  (always two taps)
//   [4*84, 8*84)    per destination row dy:    {first source row, beta0, beta1, beta2}   (beta2 = 0 for two)
#define SF_TAB_WORDS (8 * SF_OUT)
// Device copies of the static background: variant v (bit 0: score 0000000 baked in, bit 1: empty bar
// baked in) at byte offset v * SF_BG_STRIDE (92x90) and v * 84*84 (resampled)
#define SF_BG_STRIDE 8288
// backgrounds on the device: the four variants, then the same four with the live fortress at 0, 10, ... 350 degrees in them
// (sf_bg_fort_kernel: index 4 (1 + sector) + variant)
#define SF_BG_COUNT (4 * 37)
// per-env cache of the dead ship's explosion pixels (sf_render.hip: ship_explosion)
// ... followed by the score box and the bar box as they end up when only that explosion is on them (keyed by the points /
// the bar's state): SF_XC_BYTES = 1600 + 320 + 384
#define SF_XC_BYTES 2304
// The score and the bar drawn on the bare background, one picture per value (sf_render.hip: hud pictures): the score
// for -SF_HUD_SCORE_HALF <= points < SF_HUD_SCORE_HALF, the bar for its 12 states (0..10 tenths, and the kill-ready
// white one).  A picture = the rows of the surface box, SF_HUD_*_ROW bytes apart, then the rows of the 84x84 box.
#define SF_HUD_SCORE_HALF 512
#define SF_HUD_SCORE_ROW 28
#define SF_HUD_SCORE_BYTES 320
#define SF_HUD_BAR_STATES 12
#define SF_HUD_BAR_ROW 40
#define SF_HUD_BAR_BYTES 384
#define SF_HUD_BYTES (2 * SF_HUD_SCORE_HALF * SF_HUD_SCORE_BYTES + SF_HUD_BAR_STATES * SF_HUD_BAR_BYTES)
// one fortress picture (sf_render.hip: fort_patch_copy): 16 x 16 of the surface, 18 rows x 20 of the 84x84 image
#define SF_FP_BYTES 640

namespace sfr {

// Area of (convex polygon, n <= 8 vertices, counter-clockwise or clockwise) intersected with the unit
// pixel [px, px+1] x [py, py+1]: Sutherland-Hodgman against the four pixel edges, then the shoelace
// formula.  T = float on the device, double on the host.
template <typename T>
SF_HD T clip_area(const T* vx, const T* vy, int n, T px, T py) {
  T ax[12], ay[12], bx[12], by[12];
  for (int i = 0; i < n; i++) {
    ax[i] = vx[i] - px;
    ay[i] = vy[i] - py;
  }
  int m = n;
  // clip against x >= 0, x <= 1, y >= 0, y <= 1 in turn
  for (int e = 0; e < 4; e++) {
    int k = 0;
    for (int i = 0; i < m; i++) {
      const int j = (i + 1 == m) ? 0 : i + 1;
      const T x0 = ax[i], y0 = ay[i], x1 = ax[j], y1 = ay[j];
      T d0, d1;
      if (e == 0) { d0 = x0; d1 = x1; }
      else if (e == 1) { d0 = (T)1 - x0; d1 = (T)1 - x1; }
      else if (e == 2) { d0 = y0; d1 = y1; }
      else { d0 = (T)1 - y0; d1 = (T)1 - y1; }
      const bool in0 = d0 >= 0, in1 = d1 >= 0;
      if (in0) {
        bx[k] = x0;
        by[k] = y0;
        k++;
      }
      if (in0 != in1) {
        const T t = d0 / (d0 - d1);
        bx[k] = x0 + t * (x1 - x0);
        by[k] = y0 + t * (y1 - y0);
        k++;
      }
    }
    m = k;
    if (m == 0) return (T)0;
    for (int i = 0; i < m; i++) {
      ax[i] = bx[i];
      ay[i] = by[i];
    }
  }
  T s = 0;
  for (int i = 0; i < m; i++) {
    const int j = (i + 1 == m) ? 0 : i + 1;
    s += ax[i] * ay[j] - ax[j] * ay[i];
  }
  s = s < 0 ? -s : s;
  return (T)0.5 * s;
}

// pixman's 8-bit multiply: round(a * b / 255)
SF_HD int mul_un8(int a, int b) {
  const int t = a * b + 128;
  return (t + (t >> 8)) >> 8;
}
// OVER of a solid grey `c` through coverage `m` (both 0..255) onto destination `d`
SF_HD int over_un8(int d, int c, int m) { return mul_un8(c, m) + mul_un8(d, 255 - m); }

SF_HD int cover_to_mask(float area) { return (int)(fminf(area, 1.f) * 255.f + 0.5f); }

// user space -> device space (cairo_scale(.2), cairo_translate(-130, -80): SRC/draw.cpp:259-260), float32
SF_HD float dev_x(float x) { return (x - (float)SF_VP_X) * (float)SF_SCALE; }
SF_HD float dev_y(float y) { return (y - (float)SF_VP_Y) * (float)SF_SCALE; }

// ---- score text and vulnerability bar: axis-aligned rectangles, evaluated per pixel.  Shared by the
// kernel and by the host, which bakes the two most common cases (score 0000000, empty bar) into
// variants of the static background with exactly this arithmetic.

// pixel boxes [x0, x1) x [y0, y1) that contain them
#define SF_TXT_BOX_X0 32
#define SF_TXT_BOX_X1 58
#define SF_TXT_BOX_Y0 1
#define SF_TXT_BOX_Y1 6
#define SF_BAR_BOX_X0 25
#define SF_BAR_BOX_X1 65
#define SF_BAR_BOX_Y0 88
#define SF_BAR_BOX_Y1 91

// seven-segment masks (bit 0 = A top, clockwise, bit 6 = G middle) for 0-9 and '-'
SF_HD unsigned seg_mask(int glyph) {
  constexpr unsigned char k[11] = {0x3F, 0x06, 0x5B, 0x4F, 0x66, 0x6D, 0x7D, 0x07, 0x7F, 0x6F, 0x40};
  return k[glyph];
}

// "%07d" of the score (drawScore, SRC/draw.cpp:190-203) as 7 x 7 segment bits, cell 0 = leftmost character
SF_HD unsigned long long score_masks(int pnts) {
  const bool neg = pnts < 0;
  unsigned mag = neg ? (unsigned)(-(long long)pnts) : (unsigned)pnts;
  unsigned long long masks = 0;
  for (int cell = 6; cell >= 0; cell--) {
    masks |= (unsigned long long)seg_mask((int)(mag % 10u)) << (7 * cell);
    mag /= 10u;
  }
  if (neg) masks = (masks & ~0x7Full) | seg_mask(10);
  return masks;
}

// grey .5 text over destination value d of pixel (px, py): the (at most two) glyph cells over the
// pixel, segments in A..G order -- the order a stroke-by-stroke pass would give
SF_HD int text_pixel(int px, int py, unsigned long long masks, int d) {
  const float W = SF_TXT_W, H = SF_TXT_H, T = SF_TXT_T, m0 = 0.5f * (SF_TXT_H - SF_TXT_T), m1 = 0.5f * (SF_TXT_H + SF_TXT_T);
  // segment rectangles (glyph coordinates): A, B, C, D, E, F, G
  const float sx0[7] = {0, W - T, W - T, 0, 0, 0, 0}, sx1[7] = {W, W, W, W, T, T, W};
  const float sy0[7] = {0, T, m1, H - T, m1, T, m0}, sy1[7] = {T, m0, H - T, H, H - T, m0, m1};
  const float fpx = (float)px, fpy = (float)py;
  // cells whose ink [gx, gx + W] (device: 2.8 px every 3.6 px) can touch this pixel
  int c0 = (int)floorf((fpx - dev_x(SF_TXT_X0 + SF_TXT_PAD + SF_TXT_W)) * (1.0f / (SF_TXT_ADV * (float)SF_SCALE))) + 1;
  c0 = c0 < 0 ? 0 : c0;
  for (int cell = c0; cell < c0 + 2 && cell < 7; cell++) {
    const unsigned bits = (unsigned)(masks >> (7 * cell)) & 0x7Fu;
    const float gx = SF_TXT_X0 + SF_TXT_ADV * (float)cell + SF_TXT_PAD, gy = SF_TXT_TOP;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int seg = 0; seg < 7; seg++) {
      if (!((bits >> seg) & 1u)) continue;
      const float ox = fminf(dev_x(gx + sx1[seg]), fpx + 1.f) - fmaxf(dev_x(gx + sx0[seg]), fpx);
      const float oy = fminf(dev_y(gy + sy1[seg]), fpy + 1.f) - fmaxf(dev_y(gy + sy0[seg]), fpy);
      if (ox > 0.f && oy > 0.f) {
        const int m = cover_to_mask(ox * oy);
        if (m > 0) d = over_un8(d, 128, m);
      }
    }
  }
  return d;
}

// drawVlner (SRC/draw.cpp:205-225): the .33 grey bar, then `v` (0..10) tenths of it in grey `vg`
SF_HD int bar_pixel(int px, int py, int v, int vg, int d) {
  const float bx0 = dev_x(255.f), bx1 = dev_x(455.f), by0 = dev_y(522.f), by1 = dev_y(532.f);
  const float vx1 = dev_x(255.f + 20.f * (float)v);
  const float fpx = (float)px, fpy = (float)py;
  const float oy = fmaxf(fminf(by1, fpy + 1.f) - fmaxf(by0, fpy), 0.f);
  const float o1 = fmaxf(fminf(bx1, fpx + 1.f) - fmaxf(bx0, fpx), 0.f);
  const float o2 = fmaxf(fminf(vx1, fpx + 1.f) - fmaxf(bx0, fpx), 0.f);
  const int ma = cover_to_mask(o1 * oy);
  if (ma > 0) d = over_un8(d, 84, ma);
  const int mb = cover_to_mask(o2 * oy);
  if (v > 0 && mb > 0) d = over_un8(d, vg, mb);
  return d;
}

}  // namespace sfr
