// sf_host.cpp -- host-side constants of the path: presets, action tables, the libc spawn
// sequence, the integer-degree trig table, hexagon edges.  No HIP calls in this file, so
// everything here is exercised by the CPU test-suite through the C ABI.
//
// SRC = python/spacefortress/src, ENV = python/spacefortress.gym/spacefortress/gym/envs/ssf_env.py
// of the reference.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sf_internal.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

thread_local char g_err[512] = "";

// The reference reaches libm's cos() and sin() as two separate calls (deg2rad() is out of line
// in another translation unit, SRC/vector.cpp:34-36).  glibc's sincos() is NOT bit-identical to
// sin() at 48, 297 and 342 degrees, so a compiler must never fuse these: call through volatile
// pointers.
double (*volatile libm_cos)(double) = cos;
double (*volatile libm_sin)(double) = sin;

double deg2rad(double a) { return a * M_PI / 180; }

void hex_points(int radius, double* p /* [6][2] */) {  // Hexagon::setRadius, SRC/hexagon.cpp:13-34
  double x1 = floor(355 - radius);
  double x2 = floor(355 - radius * 0.5);
  double x3 = floor(355 + radius * 0.5);
  double x4 = floor(355 + radius);
  double y1 = 315;
  double y2 = floor(315 - radius * libm_sin(M_PI * 2 / 3));
  double y3 = floor(315 + radius * libm_sin(M_PI * 2 / 3));
  const double pts[6][2] = {{x1, y1}, {x2, y2}, {x3, y2}, {x4, y1}, {x3, y3}, {x2, y3}};
  memcpy(p, pts, sizeof(pts));
}

// per edge (nx, ny, px, py) exactly as Hexagon::isInside forms them (SRC/hexagon.cpp:38-42)
void hex_edges(int radius, double* e /* [6][4] */) {
  double p[12];
  hex_points(radius, p);
  for (int i = 0; i < 6; i++) {
    int j = (i + 1) % 6;
    e[4 * i + 0] = -(p[2 * j + 1] - p[2 * i + 1]);
    e[4 * i + 1] = p[2 * j] - p[2 * i];
    e[4 * i + 2] = p[2 * i];
    e[4 * i + 3] = p[2 * i + 1];
  }
}

bool edges_inside(const double* e, double x, double y) {  // SRC/hexagon.cpp:36-48
  for (int i = 0; i < 6; i++) {
    double dx = x - e[4 * i + 2], dy = y - e[4 * i + 3];
    if (e[4 * i] * dx + e[4 * i + 1] * dy < 0) return false;
  }
  return true;
}

// np.array(np.meshgrid([0,1] x k)).T.reshape(-1, k) (ENV:68-70,81-83): row r -> key bits.
// With 'xy' indexing grid c has shape (n1,n0,n2,n3) and value x_c[i_c]; after .T and reshape,
// row ((a*2+b)*2+c)*2+d reads i3=a, i2=b, i0=c, i1=d  (k=2: row c*2+d reads i0=c, i1=d).
void meshgrid_rows(int k, uint8_t* keys) {
  const int n = 1 << k;
  for (int r = 0; r < n; r++) {
    int idx[4] = {0, 0, 0, 0};
    if (k == 4) {
      idx[3] = (r >> 3) & 1;
      idx[2] = (r >> 2) & 1;
      idx[0] = (r >> 1) & 1;
      idx[1] = r & 1;
    } else {
      idx[0] = (r >> 1) & 1;
      idx[1] = r & 1;
    }
    uint8_t m = 0;
    for (int c = 0; c < k; c++)
      if (idx[c]) m |= (uint8_t)(1u << c);
    keys[r] = m;
  }
}

}  // namespace

void sf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sf_last_error(void) { return g_err; }
extern "C" int sf_version(void) { return SFMI_VERSION; }
#ifndef SFMI_BUILD_ID
#define SFMI_BUILD_ID "unstamped"
#endif
extern "C" const char* sf_build_id(void) { return SFMI_BUILD_ID; }

extern "C" int sf_preset_get(const char* gametype, sf_preset* c) {
  if (!gametype || !c) {
    sf_set_error("sf_preset_get: null argument");
    return SF_ERR_ARG;
  }
  memset(c, 0, sizeof(*c));
  // baseConfig, SRC/configs.cpp:3-49
  c->width = 710;
  c->height = 626;
  c->game_time = 60000;
  c->destroy_fortress = 100;
  c->ship_death_penalty = 100;
  c->missile_penalty = 2.0;
  c->miss_penalty = 0;
  c->shell_speed = 6;
  c->shell_radius = 3;
  c->missile_speed = 20;
  c->missile_radius = 5;
  c->auto_turn = 0;
  c->sector_size = 10;
  c->lock_time = 1000;
  c->vuln_time = 250;
  c->vuln_threshold = 10;
  c->fortress_radius = 18;
  c->big_hex = 200;
  c->small_hex = 40;
  c->explode_duration = 1000;
  c->start_vx = libm_cos(deg2rad(-60));
  c->start_vy = libm_sin(deg2rad(-60));
  c->ship_radius = 10;
  c->ship_accel = 0.3;
  c->turn_speed = 6;
  c->game_time = 180000;  // every preset, SRC/configs.cpp:55,66,78,86
  c->n_keys = 4;
  if (!strcmp(gametype, "autoturn")) {  // SRC/configs.cpp:51-61
    c->auto_turn = 1;
    c->destroy_fortress = 1;
    c->ship_death_penalty = 1;
    c->missile_penalty = 0.05;
    c->shaped = 1;
    c->n_keys = 2;
  } else if (!strcmp(gametype, "youturn")) {  // :63-72
    c->destroy_fortress = 1;
    c->ship_death_penalty = 1;
    c->missile_penalty = 0.05;
    c->shaped = 1;
  } else if (!strcmp(gametype, "test-autoturn")) {  // :74-81
    c->auto_turn = 1;
    c->n_keys = 2;
  } else if (!strcmp(gametype, "test-youturn")) {  // :83-89
  } else {
    // SRC/pymodule.cpp:341
    sf_set_error("cannot initialize Game. Unknown config value: `%s'", gametype);
    return SF_ERR_PRESET;
  }
  return SF_OK;
}

extern "C" int sf_action_table(const char* gametype, int action_set, uint8_t* out) {
  sf_preset p;
  int rc = sf_preset_get(gametype, &p);
  if (rc != SF_OK) return rc;
  if (!out) {
    sf_set_error("sf_action_table: null output");
    return SF_ERR_ARG;
  }
  if (p.n_keys == 4) {  // ENV:65-78
    if (action_set == 0 || action_set == -1) {
      meshgrid_rows(4, out);
      return 16;
    }
    if (action_set == 1) {
      const uint8_t t[5] = {0, 1, 2, 4, 8};  // NOOP FIRE THRUST LEFT RIGHT
      memcpy(out, t, 5);
      return 5;
    }
  } else {  // ENV:79-89
    if (action_set == -1) {  // 16 rows of 4 columns, of which step() reads two (ENV:213-220)
      meshgrid_rows(4, out);
      for (int i = 0; i < 16; i++) out[i] &= 3;
      return 16;
    }
    if (action_set == 0) {
      meshgrid_rows(2, out);
      return 4;
    }
    if (action_set == 1) {
      const uint8_t t[3] = {0, 1, 2};  // NOOP FIRE THRUST
      memcpy(out, t, 3);
      return 3;
    }
  }
  sf_set_error("action_set must be 1, 0 or -1 (got %d)", action_set);
  return SF_ERR_ARG;
}

extern "C" int sf_hex_points(int radius, double* out) {
  if (!out || radius <= 0) {
    sf_set_error("sf_hex_points: bad argument");
    return SF_ERR_ARG;
  }
  hex_points(radius, out);
  return SF_OK;
}

// cos / sin of deg2rad(deg), 0 <= deg < 360, as the reference computes them (two libm calls on vector.cpp's deg2rad)
extern "C" int sf_trig_deg(int deg, double* cos_sin) {
  const double r = deg2rad((double)deg);
  cos_sin[0] = libm_cos(r);
  cos_sin[1] = libm_sin(r);
  return SF_OK;
}

extern "C" int sf_trig_table(double* out) {
  if (!out) {
    sf_set_error("sf_trig_table: null output");
    return SF_ERR_ARG;
  }
  for (int k = 0; k < 360; k++) {
    double r = deg2rad((double)k);
    out[2 * k] = libm_cos(r);
    out[2 * k + 1] = libm_sin(r);
  }
  return SF_OK;
}

// Game::resetShip (SRC/game.cpp:133-149) driven by glibc's own generator with a PRIVATE state
// (random_r; rand() and random() are the same TYPE_3 generator in glibc), so the process-wide
// rand() stream of the caller is left alone.
extern "C" int sf_spawn_table(uint32_t seed, int n, int16_t* out) {
  if (!out || n <= 0) {
    sf_set_error("sf_spawn_table: bad argument");
    return SF_ERR_ARG;
  }
  sf_preset p;
  sf_preset_get("youturn", &p);  // hexagon radii are the same in every preset
  double big[24], small_[24];
  hex_edges(p.big_hex, big);
  hex_edges(p.small_hex, small_);
  struct random_data rd;
  char statebuf[128];
  memset(&rd, 0, sizeof(rd));
  memset(statebuf, 0, sizeof(statebuf));
  if (initstate_r(seed, statebuf, sizeof(statebuf), &rd) != 0) {
    sf_set_error("initstate_r failed");
    return SF_ERR_ARG;
  }
  for (int i = 0; i < n; i++) {
    int32_t r;
    double x, y;
    for (;;) {
      random_r(&rd, &r);
      x = r % 380 + 170;
      random_r(&rd, &r);
      y = r % 330 + 150;
      if (edges_inside(big, x, y) && !edges_inside(small_, x, y)) break;
    }
    random_r(&rd, &r);
    out[4 * i + 0] = (int16_t)x;
    out[4 * i + 1] = (int16_t)y;
    out[4 * i + 2] = (int16_t)(r % 360);
    out[4 * i + 3] = 0;
  }
  return SF_OK;
}

void sf_host_fill_consts(const sf_preset& p, double* c) {
  sf_trig_table(c + SF_LDS_TRIG);
  hex_edges(p.big_hex, c + SF_LDS_BIGHEX);
  hex_edges(p.small_hex, c + SF_LDS_SMALLHEX);
  for (int k = 0; k < SF_ATAB_DOUBLES; k++) c[SF_CONST_ATAB + k] = k <= 16 ? atan((double)k / 16.0) : 0.0;
}
