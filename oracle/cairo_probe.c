/*
 * cairo_probe.c -- TEST INFRASTRUCTURE ONLY.  A draw-script interpreter over the REAL cairo of this image
 * (conda's libcairo 1.16.0), built by `make -C oracle cairoprobe` into oracle/_ref/libcairoprobe.so.
 *
 * The reference's image observation is whatever cairo's image backend makes of the calls in
 * SRC/draw.cpp:82-270; cairo itself is a third-party dependency that is not under /root/reference
 * (python/spacefortress/setup.py:8 links the system's).  oracle/cairo_model.c restates the parts of its
 * rasteriser that those calls reach; this probe runs the SAME script through the real library so that the
 * restatement can be checked on arbitrary strokes and fills (tests/test_cairo_model.py), not only on game frames.
 *
 * Script: doubles, one opcode then its arguments (see cairo_model.h: CM_*).
 */
#include <cairo/cairo.h>
#include <string.h>

#include "cairo_model.h"

int cp_run(const double* s, int n, int w, int h, unsigned char* out) {
  cairo_surface_t* surf = cairo_image_surface_create(CAIRO_FORMAT_RGB24, w, h);
  cairo_t* cr = cairo_create(surf);
  int i = 0, rc = 0;
  while (i < n) {
    int op = (int)s[i++];
    if (op == CM_END) break;
    switch (op) {
      case CM_SAVE: cairo_save(cr); break;
      case CM_RESTORE: cairo_restore(cr); break;
      case CM_SCALE: cairo_scale(cr, s[i], s[i + 1]); i += 2; break;
      case CM_TRANSLATE: cairo_translate(cr, s[i], s[i + 1]); i += 2; break;
      case CM_ROTATE: cairo_rotate(cr, s[i]); i += 1; break;
      case CM_LINE_WIDTH: cairo_set_line_width(cr, s[i]); i += 1; break;
      case CM_GREY: cairo_set_source_rgb(cr, s[i], s[i], s[i]); i += 1; break;
      case CM_MOVE_TO: cairo_move_to(cr, s[i], s[i + 1]); i += 2; break;
      case CM_LINE_TO: cairo_line_to(cr, s[i], s[i + 1]); i += 2; break;
      case CM_CLOSE: cairo_close_path(cr); break;
      case CM_ARC: cairo_arc(cr, s[i], s[i + 1], s[i + 2], s[i + 3], s[i + 4]); i += 5; break;
      case CM_RECT: cairo_rectangle(cr, s[i], s[i + 1], s[i + 2], s[i + 3]); i += 4; break;
      case CM_STROKE: cairo_stroke(cr); break;
      case CM_FILL: cairo_fill(cr); break;
      case CM_PAINT: cairo_paint(cr); break;
      case CM_CURVE_TO: cairo_curve_to(cr, s[i], s[i + 1], s[i + 2], s[i + 3], s[i + 4], s[i + 5]); i += 6; break;
      case CM_NEW_PATH: cairo_new_path(cr); break;
      default: rc = -1; i = n; break;
    }
  }
  if (cairo_status(cr) != CAIRO_STATUS_SUCCESS) rc = -2;
  cairo_destroy(cr);
  cairo_surface_flush(surf);
  const unsigned char* raw = cairo_image_surface_get_data(surf);
  int stride = cairo_image_surface_get_stride(surf);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) out[y * w + x] = raw[y * stride + 4 * x];
  cairo_surface_destroy(surf);
  return rc;
}

const char* cp_version(void) { return cairo_version_string(); }

/* ---- the score text -------------------------------------------------------------------------------------------------------
 * drawScore / centeredText (SRC/draw.cpp:147-173) go through cairo's toy font API: "monospace" bold, size 30 user units,
 * cairo_text_extents -> cairo_move_to(x - width / 2, y + height / 2) -> cairo_show_text.  What lands on the surface is
 * FreeType's A8 bitmap of every glyph (the font fontconfig resolves on the box; DejaVu Sans Mono Bold in this image), blitted
 * at its origin rounded to whole pixels and composited with pixman's OVER.  These two entry points run the SAME calls on an
 * A8 surface with an opaque source, so that a byte of the result IS the mask's coverage; tests/golden/frames/
 * make_score_golden.py turns them into the glyph atlas (data) the product's text is drawn from, and checks the atlas against
 * frames of the reference's own renderer. */
static cairo_t* cp_text_ctx(cairo_surface_t* surf, double sx, double sy, double vx, double vy) {
  cairo_t* cr = cairo_create(surf);
  cairo_scale(cr, sx, sy);          /* drawGameStateScaled, SRC/draw.cpp:259-260 */
  cairo_translate(cr, -vx, -vy);
  cairo_select_font_face(cr, "monospace", CAIRO_FONT_SLANT_NORMAL, CAIRO_FONT_WEIGHT_BOLD); /* :165 */
  cairo_set_font_size(cr, 30);      /* :167 */
  cairo_set_source_rgba(cr, 0, 0, 0, 1);
  return cr;
}
static void cp_a8_out(cairo_surface_t* surf, int w, int h, unsigned char* out) {
  cairo_surface_flush(surf);
  const unsigned char* raw = cairo_image_surface_get_data(surf);
  int stride = cairo_image_surface_get_stride(surf);
  for (int y = 0; y < h; y++) memcpy(out + (size_t)y * w, raw + (size_t)y * stride, (size_t)w);
}
/* the whole string as centeredText(ctx, text, cx, cy) places it; ext6 = x_bearing, y_bearing, width, height, x_advance,
 * y_advance of cairo_text_extents (user units) */
int cp_score_mask(const char* text, int w, int h, double sx, double sy, double vx, double vy, int cx, int cy,
                  unsigned char* out, double* ext6) {
  cairo_surface_t* surf = cairo_image_surface_create(CAIRO_FORMAT_A8, w, h);
  cairo_t* cr = cp_text_ctx(surf, sx, sy, vx, vy);
  cairo_text_extents_t e;
  cairo_text_extents(cr, text, &e);
  if (ext6) { ext6[0] = e.x_bearing; ext6[1] = e.y_bearing; ext6[2] = e.width; ext6[3] = e.height; ext6[4] = e.x_advance; ext6[5] = e.y_advance; }
  cairo_move_to(cr, cx - e.width / 2.0, cy + e.height / 2.0);  /* :157 */
  cairo_show_text(cr, text);
  int rc = cairo_status(cr) == CAIRO_STATUS_SUCCESS ? 0 : -2;
  cairo_destroy(cr);
  cp_a8_out(surf, w, h, out);
  cairo_surface_destroy(surf);
  return rc;
}
/* `text` with its origin at DEVICE position (dev_x, dev_y): a glyph's bitmap relative to a whole-pixel origin */
int cp_text_at(const char* text, int w, int h, double sx, double sy, double vx, double vy, double dev_x, double dev_y,
               unsigned char* out) {
  cairo_surface_t* surf = cairo_image_surface_create(CAIRO_FORMAT_A8, w, h);
  cairo_t* cr = cp_text_ctx(surf, sx, sy, vx, vy);
  double ux = dev_x, uy = dev_y;
  cairo_device_to_user(cr, &ux, &uy);
  cairo_move_to(cr, ux, uy);
  cairo_show_text(cr, text);
  int rc = cairo_status(cr) == CAIRO_STATUS_SUCCESS ? 0 : -2;
  cairo_destroy(cr);
  cp_a8_out(surf, w, h, out);
  cairo_surface_destroy(surf);
  return rc;
}
