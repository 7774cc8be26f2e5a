/* cairo_model.h -- TEST INFRASTRUCTURE ONLY: the draw-script vocabulary shared by cairo_model.c (our restatement of
 * cairo 1.16's image rasteriser) and cairo_probe.c (the same script through the real library). */
#ifndef SF_CAIRO_MODEL_H
#define SF_CAIRO_MODEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  CM_END = 0,
  CM_SAVE = 1,       /* cairo_save */
  CM_RESTORE = 2,    /* cairo_restore */
  CM_SCALE = 3,      /* sx sy */
  CM_TRANSLATE = 4,  /* tx ty */
  CM_ROTATE = 5,     /* radians */
  CM_LINE_WIDTH = 6, /* w */
  CM_GREY = 7,       /* g: cairo_set_source_rgb(g, g, g) */
  CM_MOVE_TO = 8,    /* x y */
  CM_LINE_TO = 9,    /* x y */
  CM_CLOSE = 10,
  CM_ARC = 11,  /* xc yc r a1 a2 */
  CM_RECT = 12, /* x y w h */
  CM_STROKE = 13,
  CM_FILL = 14,
  CM_PAINT = 15,
  CM_CURVE_TO = 16, /* x1 y1 x2 y2 x3 y3 */
  CM_NEW_PATH = 17
};

/* Run a script on a fresh w x h RGB24-like surface (one grey channel, starts at 0); out[h][w]. 0 = ok. */
int cm_run(const double* script, int n, int w, int h, unsigned char* out);

/* Same, but composite onto the frame already in `out` (a baked background). */
int cm_run_over(const double* script, int n, int w, int h, unsigned char* out);

/* The polygon the LAST stroke / fill of the last cm_run produced, as cairo_edge_t records (24.8 fixed):
 * x1 y1 x2 y2 top bottom dir, 7 int32 per edge; returns the edge count (copies at most cap edges). */
int cm_last_polygon(int32_t* out, int cap);

int cp_run(const double* script, int n, int w, int h, unsigned char* out); /* cairo_probe.c: the real library */
const char* cp_version(void);

#ifdef __cplusplus
}
#endif
#endif
