// sf_glyphs.h -- the score text of the image observation (drawScore / centeredText, SRC/draw.cpp:147-173), shared by the
// frame kernels and the host.
//
// The reference draws "%07d" of the points through cairo's toy font API ("monospace" bold, 30 user units).  On cairo's
// image backend that is: FreeType's 8-bit coverage bitmap of each glyph, blitted with its origin rounded to whole pixels,
// composited as a solid grey .5 (128) IN that coverage OVER the surface with pixman's arithmetic (sfr::over_un8) -- no
// rasterisation at draw time at all.  The text is therefore DATA plus a placement rule, a glyph atlas:
//   alpha[c][gh][gw]   coverage of '0'..'9' (c = 0..9) and '-' (c = 10), one gw x gh box per character
//   advance            pixels from one box to the next (cairo's hinted advance is whole in device space)
//   y0                 top row of the boxes on the surface
//   x0[first][last]    left column of the FIRST box: centeredText centres on the string's ink width, which in some
//                      geometries depends on the first and the last character
// The built-in atlas below is the default geometry's (SSF_Env(scale=.2): 6-pixel DejaVu Sans Mono Bold, what fontconfig
// resolves "monospace bold" to on this image), taken from the image's real cairo + FreeType by
// tests/golden/frames/make_score_golden.py and held to 2 520 frames of the reference's own renderer (scores.npz) -- and the
// frame fixtures' rows 0..8, which no test masks any more.  Another geometry or another box's font: sf_set_score_glyphs.
// Without an atlas (gw == 0) the text is the SEVEN-SEGMENT FALLBACK of sf_raster.h, which equals no reference pixels.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define SFG_HD __host__ __device__ __forceinline__
#else
#define SFG_HD inline
#endif

#define SF_GLYPH_CHARS 11
#define SF_GLYPH_MAX_W 24
#define SF_GLYPH_MAX_H 24

// what the kernels read (device memory, one per batch) -- also the host's working copy
struct SfGlyphAtlas {
  int32_t gw, gh, advance, y0;       // gw == 0: no atlas, the seven-segment fallback
  int16_t x0[SF_GLYPH_CHARS * 10];   // [first character][last character]
  int16_t x_min, x_max;              // min / max of x0: the text's box is [x_min, x_max + 6 advance + gw) x [y0, y0 + gh)
  uint8_t alpha[SF_GLYPH_CHARS * SF_GLYPH_MAX_W * SF_GLYPH_MAX_H];  // [c][gh][gw], dense
};

namespace sfg {

// the default geometry's atlas: layout 4 x 4 boxes every 4 pixels from column 31, rows 1..4 (score_glyphs.npz: alpha_0,
// layout_0, x0_0; tests/test_capi_host.py compares)
constexpr int kDefW = 4, kDefH = 4, kDefAdvance = 4, kDefX0 = 31, kDefY0 = 1;
static const uint8_t kDefAlpha[SF_GLYPH_CHARS * kDefW * kDefH] = {
        /* '0' */ 58,  193, 182, 3,   150, 143, 185, 50,  150, 69,  171, 49,  58,  193, 182, 3,
        /* '1' */ 67,  216, 92,  0,   0,   124, 92,  0,   0,   124, 92,  0,   79,  214, 204, 57,
        /* '2' */ 100, 165, 188, 4,   0,   11,  212, 9,   12,  175, 60,  0,   153, 211, 176, 19,
        /* '3' */ 79,  167, 199, 10,  0,   154, 185, 0,   0,   0,   181, 35,  122, 175, 190, 10,
        /* '4' */ 0,   110, 218, 0,   64,  112, 212, 0,   147, 168, 240, 57,  0,   0,   212, 0,
        /* '5' */ 112, 197, 165, 0,   108, 188, 148, 1,   1,   0,   191, 39,  100, 175, 181, 6,
        /* '6' */ 41,  186, 161, 2,   139, 191, 170, 10,  143, 96,  154, 62,  56,  193, 193, 16,
        /* '7' */ 107, 176, 243, 29,  0,   35,  188, 0,   0,   144, 81,  0,   12,  208, 3,   0,
        /* '8' */ 72,  181, 192, 10,  74,  204, 197, 1,   118, 59,  161, 42,  89,  193, 191, 12,
        /* '9' */ 84,  182, 179, 2,   163, 53,  197, 46,  69,  179, 228, 42,  66,  166, 158, 0,
        /* '-' */ 0,   0,   0,   0,   0,   0,   0,   0,   24,  196, 143, 0,   0,   0,   0,   0,
};

// "%07d" of the points as seven character indices (0..9, 10 = '-'), four bits each, character 0 (leftmost) in the low bits.
// (beyond seven characters -- 10^7 points, or -10^6 -- printf grows the string; a game ends with hundreds: the low digits)
SFG_HD uint32_t score_chars(int pnts) {
  const bool neg = pnts < 0;
  unsigned mag = neg ? (unsigned)(-(long long)pnts) : (unsigned)pnts;
  uint32_t chars = 0;
  for (int cell = 6; cell >= 0; cell--) {
    chars |= (mag % 10u) << (4 * cell);
    mag /= 10u;
  }
  if (neg) chars = (chars & ~0xFu) | 10u;
  return chars;
}

// pixman's OVER of solid grey 128 through 8-bit coverage m onto d (fast_composite_over_n_8_8888's arithmetic)
SFG_HD int over128(int d, int m) {
  int t = 128 * m + 128;
  const int s = (t + (t >> 8)) >> 8;
  t = d * (255 - m) + 128;
  return s + ((t + (t >> 8)) >> 8);
}

// the text over pixel (px, py) whose value is d: every box that holds the pixel, left to right (boxes overlap when a
// font's box is wider than its advance; their ink does not)
SFG_HD int text_pixel(const SfGlyphAtlas* A, uint32_t chars, int px, int py, int d) {
  const int gw = A->gw, gh = A->gh, adv = A->advance;
  const int ry = py - A->y0;
  if (ry < 0 || ry >= gh) return d;
  const int rx = px - A->x0[(chars & 15u) * 10 + ((chars >> 24) & 15u)];
  for (int cell = 0; cell < 7; cell++) {
    const int q = rx - cell * adv;
    if (q >= 0 && q < gw) {
      const int m = A->alpha[(((chars >> (4 * cell)) & 15u) * gh + ry) * gw + q];
      if (m) d = over128(d, m);
    }
  }
  return d;
}

}  // namespace sfg
