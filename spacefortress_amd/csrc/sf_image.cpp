// sf_image.cpp -- host-side tables of the image observation (no HIP calls; tested without a GPU):
//   * the static background: the two hexagons, which the reference strokes first on every frame
//     (SRC/draw.cpp:102-114,230-231) and which never change: a closed path with miter joins through cairo's stroker and
//     scan converter (sf_cairo_host.cpp), for any geometry;
//   * the live fortress's coverage at its 36 headings (one cairo_stroke of four lines: SRC/draw.cpp:238-242), the explosion's
//     arc constants;
//   * the INTER_AREA resampling tables of cv2.resize(frame, (84, 84)) (rl/envs.py:29).
// cv2 (OpenCV) is a dependency of the reference that is not vendored in /root/reference and not
// installed in this image; the table follows OpenCV's published algorithm (modules/imgproc/src/
// resize.cpp, computeResizeAreaTab, the general non-integer-scale area path).
#include <math.h>
#include <string.h>

#include <vector>

#include "sf_cairo_host.h"
#include "sf_drawrec.h"
#include "sf_internal.h"
#include "sf_raster.h"

// the two hexagons stroked on black for a surface of w x h pixels showing the viewport (vx, vy, vw, vh): scale(w / vw, h / vh)
// translate(-vx, -vy), line width lw user units (drawGameStateScaled, SRC/draw.cpp:256-263; newPixelBuffer :59-76)
extern "C" int sf_image_background_geom(int w, int h, double vx, double vy, double vw, double vh, double lw, uint8_t* out) {
  if (!out || w <= 0 || h <= 0 || !(vw > 0) || !(vh > 0) || !(lw > 0)) {
    sf_set_error("sf_image_background_geom: bad argument");
    return SF_ERR_ARG;
  }
  memset(out, 0, (size_t)w * h);   // cairo_paint of black, SRC/draw.cpp:262-263
  const sfh::Geometry g{w, h, (double)w / vw, (double)h / vh, vx, vy, lw};
  const int radii[2] = {200, 40};  // bigHex, smallHex (SRC/configs.cpp:34-35), drawn in this order
  for (int k = 0; k < 2; k++) {
    double p[12];
    sf_hex_points(radii[k], p);
    sfh::stroke_hexagon(p, g, 255, out);  // white, SRC/draw.cpp:104-107
  }
  return SF_OK;
}

// the live fortress (SRC/draw.cpp:238-242: translate(355, 315) rotate(heading), four lines, ONE stroke) over the 16 x 16 box
// the frame kernel keeps for it (sf_drawrec.h: kFpX0 ..), as 8-bit coverage, sector = heading / 10.  Heading 0 is the one
// pose whose path is rectilinear under a matrix without rotation: cairo strokes that as boxes
// (cairo-path-stroke-boxes.c: _cairo_rectilinear_stroker) through the box converter, exact area instead of sub-rows.
extern "C" int sf_image_fort_alpha(int sector, uint8_t* out256) {
  if (sector < 0 || sector > 35 || !out256) {
    sf_set_error("sf_image_fort_alpha: sector 0..35, out non-null");
    return SF_ERR_ARG;
  }
  const sfh::Geometry g{SF_IMG_W, SF_IMG_H, (double)SF_IMG_W / 450.0, (double)SF_IMG_H / 460.0, SF_VP_X, SF_VP_Y, SF_LINE_W};
  std::vector<uint8_t> a((size_t)SF_IMG_W * SF_IMG_H, 0);
  if (sector == 0) {
    static const double L[4][4] = {{0, 0, 36, 0}, {0, -18, 18, -18}, {18, -18, 18, 18}, {18, 18, 0, 18}};
    const sft::Affine v = sft::view_matrix(g.sx, g.sy, g.vp_x, g.vp_y);
    const sft::Affine m = sft::object_matrix(v, 355.0, 315.0, 1.0, 0.0);
    const int hx = sft::fx_from_double(fabs(m.xx) * g.lw / 2.0), hy = sft::fx_from_double(fabs(m.yy) * g.lw / 2.0);
    sfh::Box4 bx[4];
    for (int k = 0; k < 4; k++) {
      int x1, y1, x2, y2;
      sft::to_device(m, L[k][0], L[k][1], &x1, &y1);
      sft::to_device(m, L[k][2], L[k][3], &x2, &y2);
      if (y1 == y2) { y1 -= hy; y2 += hy; } else { x1 -= hx; x2 += hx; }
      bx[k] = sfh::Box4{x1 < x2 ? x1 : x2, y1 < y2 ? y1 : y2, x1 < x2 ? x2 : x1, y1 < y2 ? y2 : y1};
    }
    sfh::boxes_cover(bx, 4, g.w, g.h, 255, a.data());  // white on black: the pixel IS the alpha
  } else {
    double cs[2];
    sf_trig_deg(10 * sector, cs);
    const sfh::Object ob = sfh::wireframe_object(1, 355.0, 315.0, 10 * sector, g, cs);
    std::vector<int> acc((size_t)g.w * g.h, 0);
    sfh::object_coverage(ob, g.w, g.h, acc.data());
    for (int i = 0; i < g.w * g.h; i++) a[i] = (uint8_t)sft::area_to_alpha(acc[i]);
  }
  for (int y = 0; y < 16; y++)
    for (int x = 0; x < 16; x++) out256[16 * y + x] = a[(sfd::kFpY0 + y) * SF_IMG_W + sfd::kFpX0 + x];
  // nothing of it may lie outside the box
  for (int y = 0; y < SF_IMG_H; y++)
    for (int x = 0; x < SF_IMG_W; x++)
      if (a[y * SF_IMG_W + x] && !(x >= sfd::kFpX0 && x < sfd::kFpX0 + 16 && y >= sfd::kFpY0 && y < sfd::kFpY0 + 16)) {
        sf_set_error("sf_image_fort_alpha: the fortress leaves its box");
        return SF_ERR_ARG;
      }
  return SF_OK;
}

// drawExplosion's arcs (SRC/draw.cpp:116-145) as sft::ArcK, 8 doubles each: [12 ring + k] for ring 0..6, then the circle's two
// halves; made with this host's libm, which is the reference's
extern "C" int sf_arc_table(double* out /* 86 x 8 */) {
  if (!out) {
    sf_set_error("sf_arc_table: null output");
    return SF_ERR_ARG;
  }
  const double kPi = 3.14159265358979323846;
  int n = 0, ofs = 0;
  auto put = [&](const sft::ArcK& k) {
    const double v[8] = {k.rca, k.rsa, k.hrsa, k.hrca, k.rcb, k.rsb, k.hrsb, k.hrcb};
    memcpy(out + 8 * n++, v, sizeof(v));
  };
  for (int radius = 15; radius < 70; radius += 8) {
    ofs += 3;
    for (int angle = 0; angle < 360; angle += 30) put(sft::arc_k((double)radius, (angle + ofs) * kPi / 180, (angle + ofs + 10) * kPi / 180));
  }
  const double mid = 0.0 + (2 * kPi - 0.0) / 2.0;  // cairo halves an arc longer than pi (_cairo_arc_in_direction)
  put(sft::arc_k(7.0, 0.0, mid));
  put(sft::arc_k(7.0, mid, 2 * kPi));
  return SF_OK;
}

extern "C" int sf_image_background(uint8_t* out) {
  if (!out) {
    sf_set_error("sf_image_background: null output");
    return SF_ERR_ARG;
  }
  // (user space -> device space: cairo_scale(90 / 450, 92 / 460) then cairo_translate(-130, -80), SRC/draw.cpp:259-260)
  return sf_image_background_geom(SF_IMG_W, SF_IMG_H, SF_VP_X, SF_VP_Y, 450.0, 460.0, SF_LINE_W, out);
}

extern "C" int sf_resize_area_tab(int ssize, int dsize, int32_t* first, int32_t* count, float* alpha) {
  // (the default geometry shrinks by 15/14 and 23/21: two or three taps; any geometry below a threefold shrink has at most
  //  four: floor(scale) + 2)
  if (ssize <= 0 || dsize <= 0 || dsize > ssize || ssize >= 3 * dsize || !first || !count || !alpha) {
    sf_set_error("sf_resize_area_tab: need dsize <= ssize < 3*dsize and non-null outputs");
    return SF_ERR_ARG;
  }
  const double inv_scale = (double)dsize / (double)ssize;
  const double scale = 1.0 / inv_scale;
  for (int dx = 0; dx < dsize; dx++) {
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = fmin(scale, ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    if (sx2 > ssize - 1) sx2 = ssize - 1;
    if (sx1 > sx2) sx1 = sx2;
    int k = 0, f = sx1;
    float* a = alpha + 4 * dx;
    a[0] = a[1] = a[2] = a[3] = 0.f;
    if (sx1 - fsx1 > 1e-3) {
      f = sx1 - 1;
      a[k++] = (float)((sx1 - fsx1) / cell);
    }
    for (int sx = sx1; sx < sx2; sx++) a[k++] = (float)(1.0 / cell);
    if (fsx2 - sx2 > 1e-3) a[k++] = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
    first[dx] = f;
    count[dx] = k;  // <= 3 for ssize < 2*dsize, <= 4 for ssize < 3*dsize
  }
  return SF_OK;
}

extern "C" int sf_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  // OpenCV's resizeArea_<uchar, float>: for every row-table entry (dy, sy, beta) the source row is first
  // reduced horizontally, buf[dx] = sum alpha * S[sx] in column-table order, then sum[dx] (+)= beta * buf[dx];
  // a destination row is written when the next one starts: saturate_cast<uchar>(sum) = round half to even.
  if (!src || !dst || dw > 4096 || dh > 4096) {
    sf_set_error("sf_resize_area_u8: bad argument");
    return SF_ERR_ARG;
  }
  std::vector<int32_t> xf(dw), xc(dw), yf(dh), yc(dh);
  std::vector<float> xa(4 * (size_t)dw), ya(4 * (size_t)dh);
  int rc = sf_resize_area_tab(sw, dw, xf.data(), xc.data(), xa.data());
  if (rc != SF_OK) return rc;
  rc = sf_resize_area_tab(sh, dh, yf.data(), yc.data(), ya.data());
  if (rc != SF_OK) return rc;
  std::vector<float> buf(dw), sum(dw);
  for (int dy = 0; dy < dh; dy++) {
    for (int k = 0; k < yc[dy]; k++) {
      const uint8_t* S = src + (size_t)(yf[dy] + k) * sw;
      const float beta = ya[4 * dy + k];
      for (int dx = 0; dx < dw; dx++) {
        float b = 0.f;
        for (int j = 0; j < xc[dx]; j++) b += (float)S[xf[dx] + j] * xa[4 * dx + j];
        buf[dx] = b;
      }
      for (int dx = 0; dx < dw; dx++) sum[dx] = k == 0 ? beta * buf[dx] : sum[dx] + beta * buf[dx];
    }
    for (int dx = 0; dx < dw; dx++) {
      float r = nearbyintf(sum[dx]);
      dst[(size_t)dy * dw + dx] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
    }
  }
  return SF_OK;
}

// the built-in glyph atlas of the default geometry (sf_glyphs.h)
void sf_glyphs_default(SfGlyphAtlas* G) {
  memset(G, 0, sizeof(*G));
  G->gw = sfg::kDefW;
  G->gh = sfg::kDefH;
  G->advance = sfg::kDefAdvance;
  G->y0 = sfg::kDefY0;
  for (int i = 0; i < SF_GLYPH_CHARS * 10; i++) G->x0[i] = sfg::kDefX0;
  G->x_min = G->x_max = sfg::kDefX0;
  memcpy(G->alpha, sfg::kDefAlpha, sizeof(sfg::kDefAlpha));
}

// caller's layout + dense alpha -> the kernels' atlas; layout null = no atlas (the seven-segment fallback).  `box`: the
// pixel box the text must stay in (the default geometry's fast kernel keeps a fixed one), or null
int sf_glyphs_pack(const sf_score_glyphs* layout, const uint8_t* alpha, const int* box, SfGlyphAtlas* G) {
  memset(G, 0, sizeof(*G));
  if (!layout) return SF_OK;
  if (!alpha || layout->gw < 1 || layout->gw > SF_GLYPH_MAX_W || layout->gh < 1 || layout->gh > SF_GLYPH_MAX_H || layout->advance < 1 ||
      layout->advance > 64) {
    sf_set_error("score glyphs: need alpha and 1 <= gw <= %d, 1 <= gh <= %d, 1 <= advance <= 64", SF_GLYPH_MAX_W, SF_GLYPH_MAX_H);
    return SF_ERR_ARG;
  }
  G->gw = layout->gw;
  G->gh = layout->gh;
  G->advance = layout->advance;
  G->y0 = layout->y0;
  int lo = 32767, hi = -32768;
  for (int i = 0; i < SF_GLYPH_CHARS * 10; i++) {
    const int v = layout->x0[i / 10][i % 10];
    G->x0[i] = (int16_t)v;
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  G->x_min = (int16_t)lo;
  G->x_max = (int16_t)hi;
  memcpy(G->alpha, alpha, (size_t)SF_GLYPH_CHARS * layout->gw * layout->gh);
  if (box) {
    // every inked pixel of every character in every cell, for every placement, inside [box[0], box[2]) x [box[1], box[3])
    for (int c = 0; c < SF_GLYPH_CHARS; c++)
      for (int r = 0; r < G->gh; r++)
        for (int q = 0; q < G->gw; q++)
          if (G->alpha[(c * G->gh + r) * G->gw + q]) {
            const int y = G->y0 + r, xa = lo + q, xb = hi + 6 * G->advance + q;
            if (y < box[1] || y >= box[3] || xa < box[0] || xb >= box[2]) {
              sf_set_error("score glyphs: ink at row %d, columns %d..%d leaves the text box [%d, %d) x [%d, %d) of the default geometry",
                           y, xa, xb, box[0], box[2], box[1], box[3]);
              return SF_ERR_ARG;
            }
          }
  }
  return SF_OK;
}

int sf_image_static_glyphs(int variant, const SfGlyphAtlas* G, uint8_t* out) {
  // the static background plus what the kernel may take as given (sf_render.hip): bit 0 the score text
  // "0000000", bit 1 the vulnerability bar at 0 -- drawn with the kernel's own per-pixel arithmetic
  if (variant < 0 || variant > 3 || !out) {
    sf_set_error("sf_image_static: variant must be 0..3 and out non-null");
    return SF_ERR_ARG;
  }
  int rc = sf_image_background(out);
  if (rc != SF_OK) return rc;
  if (variant & 1) {
    const unsigned long long masks = sfr::score_masks(0);
    const uint32_t chars = sfg::score_chars(0);
    for (int y = SF_TXT_BOX_Y0; y < SF_TXT_BOX_Y1; y++)
      for (int x = SF_TXT_BOX_X0; x < SF_TXT_BOX_X1; x++) {
        uint8_t* p = out + y * SF_IMG_W + x;
        *p = (uint8_t)(G && G->gw ? sfg::text_pixel(G, chars, x, y, *p) : sfr::text_pixel(x, y, masks, *p));
      }
  }
  if (variant & 2)
    for (int y = SF_BAR_BOX_Y0; y < SF_BAR_BOX_Y1; y++)
      for (int x = SF_BAR_BOX_X0; x < SF_BAR_BOX_X1; x++)
        out[y * SF_IMG_W + x] = (uint8_t)sfr::bar_pixel(x, y, 0, 168, out[y * SF_IMG_W + x]);
  return SF_OK;
}

extern "C" int sf_image_static(int variant, uint8_t* out) {
  SfGlyphAtlas G;
  sf_glyphs_default(&G);
  return sf_image_static_glyphs(variant, &G, out);
}

extern "C" int sf_default_score_glyphs(sf_score_glyphs* layout, uint8_t* alpha, size_t alpha_bytes) {
  if (!layout || !alpha || alpha_bytes < sizeof(sfg::kDefAlpha)) {
    sf_set_error("sf_default_score_glyphs: need layout and %zu bytes for alpha", sizeof(sfg::kDefAlpha));
    return SF_ERR_ARG;
  }
  layout->gw = sfg::kDefW;
  layout->gh = sfg::kDefH;
  layout->advance = sfg::kDefAdvance;
  layout->y0 = sfg::kDefY0;
  for (int i = 0; i < SF_GLYPH_CHARS * 10; i++) layout->x0[i / 10][i % 10] = sfg::kDefX0;
  memcpy(alpha, sfg::kDefAlpha, sizeof(sfg::kDefAlpha));
  return SF_OK;
}
