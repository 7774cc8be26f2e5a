// sf_raster.h -- geometry of the image observation, shared by the HIP render kernel and the host
// (static background, resize tables).  What is drawn, where, in which order and colour follows the
// reference renderer (SRC/draw.cpp:227-270, SRC/wireframe.cpp:8-70, ENV:50-58,203-206,
// rl/envs.py:28-30); HOW a stroke becomes pixel coverage is cairo's business and cairo is not in
// this image, so the anti-aliasing here is our own model -- exact area coverage of the stroke
// rectangles -- and pixel parity with cairo + cv2 is UNPINNED ("image observation").
#pragma once

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
// (355, 97), grey .5.  No font rasteriser here: seven-segment glyphs in a cell of about the same
// metrics (advance .6 em = 18, cap height .73 em = 22, stem 5), user units.  At .2 scale a glyph is
// 2.8 x 4.4 pixels either way.
#define SF_TXT_ADV 18.0f
#define SF_TXT_X0 (355.0f - 3.5f * SF_TXT_ADV)
#define SF_TXT_PAD 2.0f
#define SF_TXT_W 14.0f
#define SF_TXT_H 22.0f
#define SF_TXT_T 5.0f
#define SF_TXT_TOP (97.0f - 0.5f * SF_TXT_H)

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

}  // namespace sfr
