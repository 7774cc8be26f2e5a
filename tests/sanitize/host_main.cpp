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
  int32_t f[84], c[84];
  float a[84 * 4];
  if (sf_resize_area_tab(90, 84, f, c, a) || sf_resize_area_tab(92, 84, f, c, a)) return 6;
  printf("ok %s\n", sf_last_error());
  return 0;
}
