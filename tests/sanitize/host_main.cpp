// Host-only entry points of libsfmi (sf_host.cpp, sf_image.cpp: no HIP calls) under -fsanitize=address,undefined
// (tests/test_sanitizers.py).  GPU sanitizers are not available on the pool; these are the CPU-side ones.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "sfmi.h"
int main() {
  std::vector<int16_t> sp(4 * 65536);
  if (sf_spawn_table(1, 65536, sp.data())) return 1;
  std::vector<double> trig(720);
  if (sf_trig_table(trig.data())) return 2;
  double hp[12];
  if (sf_hex_points(200, hp) || sf_hex_points(40, hp)) return 3;
  std::vector<uint8_t> fr(92 * 90), out(84 * 84);
  for (int v = 0; v < 4; v++) {
    if (sf_image_static(v, fr.data())) return 4;
    if (sf_resize_area_u8(fr.data(), 90, 92, out.data(), 84, 84)) return 5;
  }
  const char* names[] = {"youturn", "autoturn", "test-youturn", "test-autoturn", "bogus"};
  for (const char* n : names) {
    sf_preset p;
    uint8_t keys[16];
    int rc = sf_preset_get(n, &p);
    for (int as = -1; as <= 2; as++) sf_action_table(n, as, keys);
    printf("%s %d\n", n, rc);
  }
  // the image observation's host rasteriser (sf_cairo_host.cpp, sf_tor.h): every picture table and the test hooks, incl. objects
  // across the surface's borders and another geometry
  std::vector<uint8_t> fa(256), big(200 * 200);
  for (int k = 0; k < 36; k++)
    if (sf_image_fort_alpha(k, fa.data())) return 7;
  std::vector<double> arcs(86 * 8);
  if (sf_arc_table(arcs.data())) return 8;
  if (sf_image_background_geom(112, 115, 130, 80, 450, 460, 3.0, big.data())) return 9;
  for (int k = 0; k < 4; k++)
    for (int i = 0; i < 40; i++) {
      const double x = 125 + 11.7 * i, y = (i & 1) ? 78.3 + 11.9 * i : 541.2 - 3.3 * i;
      if (sf_image_object_alpha(k, x, y, (37 * i) % 360, 90, 92, 130, 80, 450, 460, 3.0, fr.data())) return 10;
      if (sf_image_object_alpha(k, x, y, (37 * i) % 360, 180, 184, 130, 80, 450, 460, 3.0, big.data())) return 11;
    }
  for (int i = 0; i < 6; i++) {
    if (sf_image_explosion_host(131.5 + 80 * i, 85.0 + 70 * i, 90, 92, 130, 80, 450, 460, 3.0, fr.data())) return 12;
    if (sf_image_explosion_host(131.5 + 80 * i, 85.0 + 70 * i, 180, 184, 130, 80, 450, 460, 3.0, big.data())) return 13;
    if (sf_image_arc_alpha(300.0, 300.0, 20.0 + 9 * i, 0.3 * i, 0.3 * i + 1.2, 180, 184, 130, 80, 450, 460, 3.0, big.data()) < 1) return 14;
  }
  int32_t f[84], c[84];
  float a[84 * 4];
  if (sf_resize_area_tab(90, 84, f, c, a) || sf_resize_area_tab(92, 84, f, c, a)) return 6;
  printf("ok %s\n", sf_last_error());
  return 0;
}
