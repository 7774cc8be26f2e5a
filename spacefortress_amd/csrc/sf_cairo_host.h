// sf_cairo_host.h -- host side of the image observation's rasterisation (sf_cairo_host.cpp)
#pragma once
#include <stdint.h>

#include <vector>

#include "sf_tor.h"

namespace sfh {

struct Geometry {  // SSF_Env(scale, viewport, ls): newPixelBuffer(w, h, viewport, lw) (SRC/draw.cpp:59-76, ENV:50-60)
  int w, h;
  double sx, sy, vp_x, vp_y, lw;  // sx = w / vp_w, sy = h / vp_h (NOT the caller's scale when vp_w * scale is not whole)
};
struct Box4 { int x1, y1, x2, y2; };  // 24.8 fixed

// an object of one cairo_stroke: convex quads, united through the signed sources (sf_tor.h)
struct Object {
  int nq;
  bool chain = false;  // the quads are consecutive pieces of one flattened curve: the faces between them are no polygon edges
  sft::Quad q[32];
  int nsrc;
  unsigned src_members[40];
  int src_sign[40];
};

void stroke_hexagon(const double* pts12, const Geometry& g, int grey, uint8_t* fb);
void boxes_cover(const Box4* bx, int nb, int W, int H, int grey, uint8_t* fb);
Box4 user_rect(const Geometry& g, double x, double y, double w, double h);
void object_coverage(const Object& ob, int W, int H, int* acc);
void all_subsets(Object* ob);
void draw_explosion(double x, double y, const Geometry& g, uint8_t* fb);
// kind: 0 ship, 1 fortress, 2 missile, 3 shell (SRC/wireframe.cpp:11-67); cos_sin = {cos, sin} of deg2rad(heading)
Object wireframe_object(int kind, double px, double py, int angle_deg, const Geometry& g, const double* cos_sin);

}  // namespace sfh
