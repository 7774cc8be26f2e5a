// sf_raster.h -- geometry of the image observation, shared by the HIP render kernel and the host
// (static background, resize tables).  What is drawn, where, in which order and colour follows the
// reference renderer (SRC/draw.cpp:227-270, SRC/wireframe.cpp:8-70, ENV:50-58,203-206,
// rl/envs.py:28-30); HOW a stroke or a filled rectangle becomes pixel values is cairo's image backend,
// restated in sf_tor.h and pinned to the reference's own frames (tests/golden/frames).  The score text is a glyph
// atlas (sf_glyphs.h: FreeType's bitmaps as the image's cairo blits them, held to the reference's frames); the
// seven-segment glyphs below are the named FALLBACK for a geometry / font without an atlas.
#pragma once
#include <math.h>

#include "sf_tor.h"

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

// Score text FALLBACK (drawScore, SRC/draw.cpp:161-173): "%07d", bold monospace 30 user units, centred on
// (355, 97), grey .5 -- used only where no glyph atlas is set (sf_glyphs.h).  Seven-segment glyphs in a cell of the same metrics,
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
// (the built-in atlas's ink: columns 31..58, rows 1..4; the fallback's: 32..57, 1..5; an atlas given for the default
//  geometry must keep its ink inside: sf_set_score_glyphs)
#define SF_TXT_BOX_X0 31
#define SF_TXT_BOX_X1 59
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

// "%07d" of the score (drawScore, SRC/draw.cpp:161-173) as 7 x 7 segment bits, cell 0 = leftmost character
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

// drawVlner (SRC/draw.cpp:207-225): the .33 grey bar, then `v` (0..10) tenths of it in grey `vg`: two cairo_rectangle +
// cairo_fill -> pixel-unaligned boxes -> cairo-rectangular-scan-converter.c: the box's exact area inside the pixel in
// 1/65536, c = area >> 8, alpha = c - (c >> 8), then the image compositor's lerp (sf_tor.h).  In the default geometry the
// rectangle's corners in 24.8 fixed point are compile-time facts: x = (255 - 130) * .2 = 25 exactly, widths 40 and 4 v
// pixels exactly, y = fixed(88.4) = 22630 / 256 and 2 pixels down (cairo adds the rounded HEIGHT to the rounded corner).
SF_HD int bar_pixel(int px, int py, int v, int vg, int d) {
  constexpr int bx0 = 25 * 256, bx1 = 65 * 256, by0 = 22630, by1 = 22630 + 512;
  const int vx1 = bx0 + 1024 * v;
  const int X0 = px * 256, Y0 = py * 256;
  const int oy = (by1 < Y0 + 256 ? by1 : Y0 + 256) - (by0 > Y0 ? by0 : Y0);
  if (oy <= 0) return d;
  auto ov = [&](int x1) { const int o = (x1 < X0 + 256 ? x1 : X0 + 256) - (bx0 > X0 ? bx0 : X0); return o > 0 ? o : 0; };
  const int ca = (ov(bx1) * oy) >> 8, cb = (ov(vx1) * oy) >> 8;
  const int aa = ca - (ca >> 8), ab = cb - (cb >> 8);
  if (aa > 0) d = sft::lerp8(84, aa, d);
  if (v > 0 && ab > 0) d = sft::lerp8(vg, ab, d);
  return d;
}

}  // namespace sfr
