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
