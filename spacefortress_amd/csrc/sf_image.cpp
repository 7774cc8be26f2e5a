// sf_image.cpp -- host-side tables of the image observation (no HIP calls; tested without a GPU):
//   * the static background: the two hexagons, which the reference strokes first on every frame
//     (SRC/draw.cpp:131-143,230-231) and which never change, as 8-bit coverage of a 92x90 surface;
//   * the INTER_AREA resampling tables of cv2.resize(frame, (84, 84)) (rl/envs.py:29).
// cv2 (OpenCV) is a dependency of the reference that is not vendored in /root/reference and not
// installed in this image; the table follows OpenCV's published algorithm (modules/imgproc/src/
// resize.cpp, computeResizeAreaTab, the general non-integer-scale area path).  See sf_raster.h for
// what is and is not pinned about pixel values.
#include <math.h>
#include <string.h>

#include <vector>

#include "sf_internal.h"
#include "sf_raster.h"

namespace {

// vertices of the polygon whose edges are the hexagon's edges moved by `off` along their outward
// normals: the outline (off > 0) / inline (off < 0) of a closed stroke with miter joins
void offset_polygon(const double* p /* [6][2] */, double off, double* qx, double* qy) {
  double area2 = 0;
  for (int i = 0; i < 6; i++) {
    const int j = (i + 1) % 6;
    area2 += p[2 * i] * p[2 * j + 1] - p[2 * j] * p[2 * i + 1];
  }
  const double orient = area2 > 0 ? 1.0 : -1.0;
  // edge i: from p[i] to p[i+1]; outward unit normal n_i; offset line: n_i . x = n_i . p[i] + off
  double nx[6], ny[6], c[6];
  for (int i = 0; i < 6; i++) {
    const int j = (i + 1) % 6;
    const double ex = p[2 * j] - p[2 * i], ey = p[2 * j + 1] - p[2 * i + 1];
    const double len = sqrt(ex * ex + ey * ey);
    nx[i] = orient * ey / len;
    ny[i] = -orient * ex / len;
    c[i] = nx[i] * p[2 * i] + ny[i] * p[2 * i + 1] + off;
  }
  // vertex i of the offset polygon = intersection of offset edges i-1 and i
  for (int i = 0; i < 6; i++) {
    const int h = (i + 5) % 6;
    const double det = nx[h] * ny[i] - ny[h] * nx[i];
    qx[i] = (c[h] * ny[i] - ny[h] * c[i]) / det;
    qy[i] = (nx[h] * c[i] - c[h] * nx[i]) / det;
  }
}

}  // namespace

// the two hexagons stroked on black for a surface of w x h pixels under scale(s) translate(-vx, -vy), line width lw user units
// (drawGameStateScaled, SRC/draw.cpp:256-263; Game(width, height, viewport, lw), SRC/pymodule.cpp:319-354)
extern "C" int sf_image_background_geom(double scale, double vx, double vy, int w, int h, double lw, uint8_t* out) {
  if (!out || w <= 0 || h <= 0 || !(scale > 0) || !(lw > 0)) {
    sf_set_error("sf_image_background_geom: bad argument");
    return SF_ERR_ARG;
  }
  memset(out, 0, (size_t)w * h);   // cairo_paint of black, SRC/draw.cpp:262-263
  const int radii[2] = {200, 40};  // bigHex, smallHex (SRC/configs.cpp:34-35), drawn in this order
  for (int k = 0; k < 2; k++) {
    double p[12], ox[6], oy[6], ix[6], iy[6];
    sf_hex_points(radii[k], p);
    offset_polygon(p, lw / 2, ox, oy);
    offset_polygon(p, -lw / 2, ix, iy);
    for (int i = 0; i < 6; i++) {
      ox[i] = (ox[i] - vx) * scale;
      oy[i] = (oy[i] - vy) * scale;
      ix[i] = (ix[i] - vx) * scale;
      iy[i] = (iy[i] - vy) * scale;
    }
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        // ring = outline minus inline; both convex, the inline inside the outline
        double a = sfr::clip_area<double>(ox, oy, 6, (double)x, (double)y) -
                   sfr::clip_area<double>(ix, iy, 6, (double)x, (double)y);
        if (a <= 0) continue;
        if (a > 1) a = 1;
        const int m = (int)(a * 255.0 + 0.5);
        out[(size_t)y * w + x] = (uint8_t)sfr::over_un8(out[(size_t)y * w + x], 255, m);  // white, :133-136
      }
  }
  return SF_OK;
}

extern "C" int sf_image_background(uint8_t* out) {
  if (!out) {
    sf_set_error("sf_image_background: null output");
    return SF_ERR_ARG;
  }
  // (user space -> device space: cairo_scale(.2) then cairo_translate(-130, -80), SRC/draw.cpp:259-260)
  return sf_image_background_geom(SF_SCALE, SF_VP_X, SF_VP_Y, SF_IMG_W, SF_IMG_H, SF_LINE_W, out);
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

extern "C" int sf_image_static(int variant, uint8_t* out) {
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
    for (int y = SF_TXT_BOX_Y0; y < SF_TXT_BOX_Y1; y++)
      for (int x = SF_TXT_BOX_X0; x < SF_TXT_BOX_X1; x++)
        out[y * SF_IMG_W + x] = (uint8_t)sfr::text_pixel(x, y, masks, out[y * SF_IMG_W + x]);
  }
  if (variant & 2)
    for (int y = SF_BAR_BOX_Y0; y < SF_BAR_BOX_Y1; y++)
      for (int x = SF_BAR_BOX_X0; x < SF_BAR_BOX_X1; x++)
        out[y * SF_IMG_W + x] = (uint8_t)sfr::bar_pixel(x, y, 0, 168, out[y * SF_IMG_W + x]);
  return SF_OK;
}
