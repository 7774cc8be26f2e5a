/*
 * sf_oracle.c -- CPU restatement of the Space Fortress env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sf_oracle.h).  Parity status: PINNED against the
 * reference engine built into oracle/_ref/ and against tests/golden/.
 *
 * Every function cites the reference lines it restates.  "SRC/" is
 * /root/reference/python/spacefortress/src, "ENV:" is
 * /root/reference/python/spacefortress.gym/spacefortress/gym/envs/ssf_env.py.
 *
 * Build with -O2 -ffp-contract=off: the reference is built by g++ for baseline
 * x86-64 (no FMA), so products and sums must round separately here as well.
 * All libm calls (atan2, cos, sin, sqrt, pow, fmod, ceil, floor) are the same
 * ones the reference makes, on the same arguments, in the same order.
 */
#include "sf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* libc rand(): glibc 2.35 stdlib/random_r.c, TYPE_3 (degree 31, separation 3).
 * The reference calls rand() at SRC/game.cpp:137,138,148 and never calls
 * srand(), so every env process runs the seed-1 stream.  glibc is not part of
 * /root/reference; this restates its published algorithm and is pinned by
 * tests against libc rand() itself and against the reference's spawns. */

void sfo_srand(sfo_rng* s, unsigned seed) {
  int32_t word;
  int i;
  if (seed == 0) seed = 1;
  s->r[0] = (int32_t)seed;
  word = (int32_t)seed;
  for (i = 1; i < 31; i++) {
    /* word = 16807 * word % 2147483647 without overflowing 31 bits */
    long hi = word / 127773;
    long lo = word % 127773;
    word = (int32_t)(16807 * lo - 2836 * hi);
    if (word < 0) word += 2147483647;
    s->r[i] = word;
  }
  s->f = 3;
  s->b = 0;
  for (i = 0; i < 310; i++) (void)sfo_rand(s);
}

int sfo_rand(sfo_rng* s) {
  uint32_t v = (uint32_t)s->r[s->f] + (uint32_t)s->r[s->b];
  s->r[s->f] = (int32_t)v;
  if (++s->f >= 31) s->f = 0;
  if (++s->b >= 31) s->b = 0;
  return (int)(v >> 1);
}

/* ------------------------------------------------------------------ */
/* SRC/vector.cpp:30-52 */

static double vec_norm(const sfo_vec* v) { return sqrt(v->x * v->x + v->y * v->y); } /* :30-32 */
static double deg2rad(double a) { return a * M_PI / 180; }                          /* :34-36 */
static double rad2deg(double a) { return a / M_PI * 180; }                          /* :38-40 */

static double std_angle(double a) { /* :42-46 */
  if (a <= -360 || a >= 360) a = fmod(a, 360);
  if (a < 0) a += 360;
  return a;
}

static double angle_to(const sfo_vec* p1, const sfo_vec* p2) { /* :48-52 */
  double a = atan2(p2->y - p1->y, p2->x - p1->x);
  if (a < 0) a += M_PI * 2;
  return rad2deg(a);
}

/* SRC/object.cpp:12-15 */
static int collided(const sfo_obj* a, const sfo_obj* b) {
  double d = sqrt(pow(a->pos.x - b->pos.x, 2) + pow(a->pos.y - b->pos.y, 2));
  return d <= (a->radius + b->radius);
}

/* SRC/hexagon.cpp:13-34 */
static void hex_set_radius(sfo_vec* pts, int radius) {
  double x1 = floor(355 - radius);
  double x2 = floor(355 - radius * 0.5);
  double x3 = floor(355 + radius * 0.5);
  double x4 = floor(355 + radius);
  double y1 = 315;
  double y2 = floor(315 - radius * sin(M_PI * 2 / 3));
  double y3 = floor(315 + radius * sin(M_PI * 2 / 3));
  pts[0].x = x1; pts[0].y = y1;
  pts[1].x = x2; pts[1].y = y2;
  pts[2].x = x3; pts[2].y = y2;
  pts[3].x = x4; pts[3].y = y1;
  pts[4].x = x3; pts[4].y = y3;
  pts[5].x = x2; pts[5].y = y3;
}

/* SRC/hexagon.cpp:36-48 */
static int hex_inside(const sfo_vec* pts, const sfo_vec* p) {
  int i;
  for (i = 0; i < 6; i++) {
    int j = (i + 1) % 6;
    double nx = -(pts[j].y - pts[i].y);
    double ny = pts[j].x - pts[i].x;
    double dx = p->x - pts[i].x;
    double dy = p->y - pts[i].y;
    if (nx * dx + ny * dy < 0) return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------ */
/* SRC/configs.cpp:3-89 */

static void base_config(sfo_config* c) { /* :3-49 */
  memset(c, 0, sizeof(*c));
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
  c->start_vx = cos(deg2rad(-60));
  c->start_vy = sin(deg2rad(-60));
  c->ship_radius = 10;
  c->ship_accel = 0.3;
  c->turn_speed = 6;
}

int sfo_preset(const char* name, sfo_config* c) {
  base_config(c);
  c->game_time = 180000; /* all four presets, :55,66,78,86 */
  if (strcmp(name, "autoturn") == 0) { /* :51-61 */
    c->auto_turn = 1;
    c->destroy_fortress = 1;
    c->ship_death_penalty = 1;
    c->missile_penalty = 0.05;
    c->shaped = 1;
  } else if (strcmp(name, "youturn") == 0) { /* :63-72 */
    c->destroy_fortress = 1;
    c->ship_death_penalty = 1;
    c->missile_penalty = 0.05;
    c->shaped = 1;
  } else if (strcmp(name, "test-autoturn") == 0) { /* :74-81 */
    c->auto_turn = 1;
  } else if (strcmp(name, "test-youturn") == 0) { /* :83-89 */
  } else {
    return -1;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* SRC/game.cpp */

static void game_reward(sfo_game* g, float amount) { /* :97-102 */
  g->reward += amount;
  g->raw_points += amount;
  g->points += amount;
  if (g->points < 0) g->points = 0;
}

static void game_penalize(sfo_game* g, float amount) { game_reward(g, -amount); } /* :104-106 */

static void maybe_reset_key_events(sfo_game* g) { /* :90-95 */
  if (g->events_processed) {
    g->n_events = 0;
    g->events_processed = 0;
  }
}

static void push_key(sfo_game* g, int sym, int state) {
  maybe_reset_key_events(g);
  if (g->n_events < SFO_MAX_KEY_EVENTS) {
    g->events[g->n_events].sym = sym;
    g->events[g->n_events].state = state;
    g->n_events++;
  }
}

void sfo_press_key(sfo_game* g, int sym) { push_key(g, sym, 1); }   /* :108-114 */
void sfo_release_key(sfo_game* g, int sym) { push_key(g, sym, 0); } /* :116-122 */

static int outside_game_area(const sfo_game* g, const sfo_vec* p) { /* :129-131 */
  return p->x < 0 || p->x > g->cfg.width || p->y > g->cfg.height || p->y < 0;
}

static void reset_ship(sfo_game* g) { /* :133-149 */
  int again = 1;
  g->ship.alive = 1;
  while (again) {
    g->ship.pos.x = sfo_rand(g->rng) % 380 + 170;
    g->ship.pos.y = sfo_rand(g->rng) % 330 + 150;
    if (hex_inside(g->big_hex, &g->ship.pos) && !hex_inside(g->small_hex, &g->ship.pos)) again = 0;
  }
  g->ship.vel.x = g->cfg.start_vx;
  g->ship.vel.y = g->cfg.start_vy;
  g->ship.angle = sfo_rand(g->rng) % 360;
}

static void monitor_ship_respawn(sfo_game* g) { /* :151-157 */
  if (!g->ship.alive && g->ship_death_timer >= g->cfg.explode_duration) {
    reset_ship(g);
    g->fort_timer = 0;
  }
}

static void fire_shell(sfo_game* g, const sfo_vec* p, double angle) { /* :159-173 */
  int i;
  for (i = 0; i < SFO_MAX_SHELLS; i++) {
    sfo_obj* s = &g->shells[i];
    if (!s->alive) {
      s->alive = 1;
      s->radius = g->cfg.shell_radius;
      s->pos = *p;
      s->angle = angle;
      s->vel.x = g->cfg.shell_speed * cos(deg2rad(angle));
      s->vel.y = g->cfg.shell_speed * sin(deg2rad(angle));
      return;
    }
  }
}

static void fire_missile(sfo_game* g, const sfo_vec* p, double angle) { /* :175-192 */
  int i;
  if (!g->ship.alive) return;
  for (i = 0; i < SFO_MAX_MISSILES; i++) {
    sfo_obj* m = &g->missiles[i];
    if (!m->alive) {
      m->alive = 1;
      m->radius = g->cfg.missile_radius;
      m->pos = *p;
      m->angle = angle;
      m->vel.x = g->cfg.missile_speed * cos(deg2rad(angle));
      m->vel.y = g->cfg.missile_speed * sin(deg2rad(angle));
      game_penalize(g, (float)g->cfg.missile_penalty);
      return;
    }
  }
}

static void update_fortress(sfo_game* g) { /* :194-216 */
  double dx = g->ship.pos.x - g->fortress.pos.x;
  double dy = g->ship.pos.y - g->fortress.pos.y;
  double angle_to_ship = std_angle(rad2deg(atan2(dy, dx)));

  if (!g->fortress.alive && g->fort_death_timer > 1000) {
    g->fort_timer = 0;
    g->fortress.alive = 1;
  }
  if (g->ship.alive) {
    g->fortress.angle = std_angle(ceil(angle_to_ship / g->cfg.sector_size) * g->cfg.sector_size);
    if (g->fortress.angle != g->fort_last_angle) {
      g->fort_last_angle = g->fortress.angle;
      g->fort_timer = 0;
    }
    if (g->fort_timer >= g->cfg.lock_time && g->ship.alive && g->fortress.alive) {
      fire_shell(g, &g->fortress.pos, angle_to_ship);
      g->fort_timer = 0;
    }
  }
}

static void process_key_state(sfo_game* g) { /* :218-272 */
  int i;
  maybe_reset_key_events(g);
  for (i = 0; i < g->n_events; i++) {
    int sym = g->events[i].sym;
    if (g->events[i].state) {
      if (sym == SFO_LEFT_KEY && !g->left_flag) {
        g->left_flag = 1;
        g->left_timer = 0;
        g->stats.total_lefts += 1;
      } else if (sym == SFO_RIGHT_KEY && !g->right_flag) {
        g->right_flag = 1;
        g->right_timer = 0;
        g->stats.total_rights += 1;
      } else if (sym == SFO_THRUST_KEY && !g->thrust_flag) {
        g->thrust_flag = 1;
        g->thrust_timer = 0;
        g->stats.total_thrusts += 1;
      } else if (sym == SFO_FIRE_KEY && !g->fire_flag) {
        fire_missile(g, &g->ship.pos, g->ship.angle);
        g->fire_flag = 1;
        /* :240-243 shot-interval telemetry: not on the step outputs */
        g->fire_timer = 0;
        g->stats.total_shots += 1;
      }
    } else {
      if (sym == SFO_LEFT_KEY && g->left_flag) {
        g->left_flag = 0;
        g->left_timer = 0;
      } else if (sym == SFO_RIGHT_KEY && g->right_flag) {
        g->right_flag = 0;
        g->right_timer = 0;
      } else if (sym == SFO_THRUST_KEY && g->thrust_flag) {
        g->thrust_flag = 0;
        g->thrust_timer = 0;
      } else if (sym == SFO_FIRE_KEY && g->fire_flag) {
        g->fire_flag = 0;
        g->fire_timer = 0;
      }
    }
  }
  if (g->left_flag && !g->right_flag)
    g->turn_flag = SFO_TURN_LEFT;
  else if (!g->left_flag && g->right_flag)
    g->turn_flag = SFO_TURN_RIGHT;
  else
    g->turn_flag = SFO_NO_TURN;
  g->events_processed = 1;
}

static void kill_ship(sfo_game* g) { /* :274-280 */
  if (g->ship.alive) {
    g->ship.alive = 0;
    g->ship_death_timer = 0;
    g->stats.ship_deaths += 1;
  }
}

static double norm_dist(double fdist, double big_hex, double small_hex) { /* :282-284 */
  return -1 + (fdist - small_hex) / ((big_hex - small_hex) / 2.0);
}

static double calc_vdir(const sfo_obj* ship, const sfo_obj* fortress) { /* :286-297 */
  double o, v, diff;
  if (vec_norm(&ship->vel) == 0.0) return 0.0;
  o = atan2(-(fortress->pos.y - ship->pos.y), fortress->pos.x - ship->pos.x);
  v = atan2(ship->vel.y, ship->vel.x);
  diff = v - o;
  if (diff > M_PI) diff -= M_PI * 2;
  if (diff < -M_PI) diff += M_PI * 2;
  return rad2deg(diff);
}

static double calc_aim(const sfo_obj* ship, const sfo_obj* fortress) { /* :299-305 */
  double o = atan2(ship->pos.y - fortress->pos.y, ship->pos.x - fortress->pos.x);
  o = rad2deg(o) - ship->angle + 180;
  if (o < -180) o = o + 360;
  return o;
}

static void compute_extra(sfo_game* g) { /* :307-312 */
  g->vdir = calc_vdir(&g->ship, &g->fortress);
  g->aim = calc_aim(&g->ship, &g->fortress);
  /* :310 -- the y term subtracts the ship from itself in the reference; kept */
  g->fdist = sqrt(pow(g->ship.pos.x - g->fortress.pos.x, 2) + pow(g->ship.pos.y - g->ship.pos.y, 2));
  g->ndist = norm_dist(g->fdist, g->cfg.big_hex, g->cfg.small_hex);
}

static void update_ship(sfo_game* g) { /* :314-351 */
  if (!g->ship.alive) return;
  if (g->cfg.auto_turn) {
    g->ship.angle = std_angle(ceil(angle_to(&g->ship.pos, &g->fortress.pos)));
  } else {
    if (g->turn_flag == SFO_TURN_LEFT)
      g->ship.angle = std_angle(g->ship.angle - g->cfg.turn_speed);
    else if (g->turn_flag == SFO_TURN_RIGHT)
      g->ship.angle = std_angle(g->ship.angle + g->cfg.turn_speed);
  }
  if (g->thrust_flag) {
    g->ship.vel.x += g->cfg.ship_accel * cos(deg2rad(g->ship.angle));
    g->ship.vel.y += g->cfg.ship_accel * sin(deg2rad(g->ship.angle));
  }
  g->ship.pos.x += g->ship.vel.x;
  g->ship.pos.y += g->ship.vel.y;

  compute_extra(g);

  if (!hex_inside(g->big_hex, &g->ship.pos)) {
    kill_ship(g);
    game_penalize(g, (float)g->cfg.ship_death_penalty);
    g->stats.big_hex_deaths += 1;
    g->col_big_hex = 1;
  } else if (hex_inside(g->small_hex, &g->ship.pos)) {
    kill_ship(g);
    game_penalize(g, (float)g->cfg.ship_death_penalty);
    g->stats.small_hex_deaths += 1;
    g->col_small_hex = 1;
  }
}

static void update_missiles(sfo_game* g) { /* :353-402 */
  int i;
  for (i = 0; i < SFO_MAX_MISSILES; i++) {
    sfo_obj* m = &g->missiles[i];
    if (!m->alive) continue;
    m->pos.x += m->vel.x;
    m->pos.y += m->vel.y;
    if (collided(m, &g->fortress)) {
      m->alive = 0;
      g->col_missile_fortress = 1;
      if (g->fortress.alive) {
        if (g->fort_vuln_timer >= g->cfg.vuln_time) {
          g->vlner += 1;
          g->stats.vlner_incs += 1;
          if (g->vlner > g->stats.max_vlner) g->stats.max_vlner = g->vlner;
        } else {
          if (g->vlner >= g->cfg.vuln_threshold + 1) {
            g->fortress.alive = 0;
            g->fort_death_timer = 0;
            game_reward(g, (float)(g->cfg.destroy_fortress + 0 /* mDestroyFortressExtraPoints, :47 */));
            g->stats.destroyed_fortresses += 1;
          } else {
            g->stats.resets += 1;
          }
          g->vlner = 0;
        }
        g->fort_vuln_timer = 0;
      }
    } else if (outside_game_area(g, &m->pos)) {
      m->alive = 0;
      game_penalize(g, (float)g->cfg.miss_penalty);
      g->stats.missed_shots += 1;
    }
  }
}

static void update_shells(sfo_game* g) { /* :404-423 */
  int i;
  for (i = 0; i < SFO_MAX_SHELLS; i++) {
    sfo_obj* s = &g->shells[i];
    if (!s->alive) continue;
    s->pos.x += s->vel.x;
    s->pos.y += s->vel.y;
    if (g->ship.alive && collided(s, &g->ship)) {
      g->col_shell_ship = 1;
      s->alive = 0;
      kill_ship(g);
      game_penalize(g, (float)g->cfg.ship_death_penalty);
      g->stats.shell_deaths += 1;
    } else if (outside_game_area(g, &s->pos)) {
      s->alive = 0;
    }
  }
}

static void step_timers(sfo_game* g, int ms) { /* :425-451 */
  g->tick += 1;
  g->fort_timer += ms;
  g->fort_death_timer += ms;
  g->fort_vuln_timer += ms;
  g->ship_death_timer += ms;
  g->fire_timer += g->fire_flag ? 1 : -1;
  g->thrust_timer += g->thrust_flag ? 1 : -1;
  g->left_timer += g->left_flag ? 1 : -1;
  g->right_timer += g->right_flag ? 1 : -1;
}

static void reset_tick(sfo_game* g) { /* :453-467 */
  g->col_big_hex = g->col_small_hex = g->col_missile_fortress = g->col_shell_ship = 0;
}

void sfo_game_init(sfo_game* g, const sfo_config* cfg, sfo_rng* rng) { /* :18-82 */
  memset(g, 0, sizeof(*g));
  g->cfg = *cfg;
  g->rng = rng;
  hex_set_radius(g->big_hex, cfg->big_hex);
  hex_set_radius(g->small_hex, cfg->small_hex);
  g->ship.radius = cfg->ship_radius;
  g->turn_flag = SFO_NO_TURN;
  reset_ship(g);
  g->fortress.alive = 1;
  g->fortress.radius = cfg->fortress_radius;
  g->fortress.pos.x = 355;
  g->fortress.pos.y = 315;
  g->fortress.angle = 180;
  g->fort_last_angle = 0;
  /* :78 adds vuln_time to a member the reference never initialises; a recycled
   * heap block only ever made it larger, so every hit counts as "slow": 250. */
  g->fort_vuln_timer = cfg->vuln_time;
  /* mExtra is read by reset() observations before the first tick computes it
   * (stale heap in the reference); defined here as the spawn-state extras. */
  compute_extra(g);
  reset_tick(g);
}

int sfo_step_one_tick(sfo_game* g, int ms) { /* :473-485 */
  g->reward = 0;
  g->time += ms;
  reset_tick(g);
  process_key_state(g);
  monitor_ship_respawn(g);
  update_ship(g);
  update_fortress(g);
  update_shells(g);
  update_missiles(g);
  step_timers(g, ms);
  return (int)g->reward; /* float -> int truncation, decl SRC/game.hh:138 */
}

int sfo_is_game_over(const sfo_game* g) { return g->time >= g->cfg.game_time; } /* :487-489 */

void sfo_game_snapshot(const sfo_game* g, sfo_snapshot* s) {
  int i;
  memset(s, 0, sizeof(*s));
  s->time = g->time;
  s->tick = g->tick;
  s->ship_alive = g->ship.alive;
  s->ship_x = g->ship.pos.x;
  s->ship_y = g->ship.pos.y;
  s->ship_vx = g->ship.vel.x;
  s->ship_vy = g->ship.vel.y;
  s->ship_angle = g->ship.angle;
  s->ship_death_timer = g->ship_death_timer;
  s->fire_timer = g->fire_timer;
  s->thrust_timer = g->thrust_timer;
  s->left_timer = g->left_timer;
  s->right_timer = g->right_timer;
  s->thrust_flag = g->thrust_flag;
  s->fire_flag = g->fire_flag;
  s->left_flag = g->left_flag;
  s->right_flag = g->right_flag;
  s->turn_flag = g->turn_flag;
  s->fort_alive = g->fortress.alive;
  s->fort_angle = g->fortress.angle;
  s->fort_last_angle = g->fort_last_angle;
  s->fort_timer = g->fort_timer;
  s->fort_death_timer = g->fort_death_timer;
  s->fort_vuln_timer = g->fort_vuln_timer;
  s->points = g->points;
  s->raw_points = g->raw_points;
  s->vlner = g->vlner;
  memcpy(s->stats, &g->stats, sizeof(s->stats));
  s->vdir = g->vdir;
  s->fdist = g->fdist;
  s->ndist = g->ndist;
  s->aim = g->aim;
  for (i = 0; i < SFO_MAX_MISSILES; i++) {
    s->missile_alive[i] = g->missiles[i].alive;
    s->missile_x[i] = g->missiles[i].pos.x;
    s->missile_y[i] = g->missiles[i].pos.y;
    s->missile_vx[i] = g->missiles[i].vel.x;
    s->missile_vy[i] = g->missiles[i].vel.y;
    s->missile_angle[i] = g->missiles[i].angle;
  }
  for (i = 0; i < SFO_MAX_SHELLS; i++) {
    s->shell_alive[i] = g->shells[i].alive;
    s->shell_x[i] = g->shells[i].pos.x;
    s->shell_y[i] = g->shells[i].pos.y;
    s->shell_vx[i] = g->shells[i].vel.x;
    s->shell_vy[i] = g->shells[i].vel.y;
    s->shell_angle[i] = g->shells[i].angle;
  }
  s->collisions = (g->col_big_hex ? 1 : 0) | (g->col_small_hex ? 2 : 0) |
                  (g->col_missile_fortress ? 4 : 0) | (g->col_shell_ship ? 8 : 0);
}

int sfo_snapshot_size(void) { return (int)sizeof(sfo_snapshot); }

/* ------------------------------------------------------------------ */
/* ENV: the gym wrapper */

/* np.array(np.meshgrid([0,1] x k)).T.reshape(-1,k) (ENV:68-70,81-83): row r,
 * column c.  For k = 4 the grids have shape (2,2,2,2) indexed [j,i,k2,k3] with
 * 'xy' indexing; after .T the index order is reversed, so row r = (d3,d2,d1,d0)
 * in mixed radix reads grid[c][d0?]...  Derived once with numpy and frozen here:
 *   k=4: column0 = bit2 of r... see tests/test_oracle_wrapper.py which checks
 *   these tables against numpy's own meshgrid. */
static void meshgrid_rows(int k, uint8_t* keys /* n = 2^k rows */) {
  /* meshgrid(x0..x{k-1}) with indexing='xy' returns arrays of shape
   * (n1, n0, n2, n3): G_c[i1,i0,i2,i3] = x_c[i_c].  A = stack(G) has shape
   * (k, n1,n0,n2,n3); A.T has shape (n3,n2,n0,n1,k) with
   * A.T[a,b,c,d,col] = G_col[d,c,b,a] -> i1=d, i0=c, i2=b, i3=a.
   * reshape(-1,k): row = ((a*2 + b)*2 + c)*2 + d for k=4. */
  int n = 1 << k, r, c;
  for (r = 0; r < n; r++) {
    int idx[4] = {0, 0, 0, 0};
    if (k == 4) {
      int a = (r >> 3) & 1, b = (r >> 2) & 1, cc = (r >> 1) & 1, d = r & 1;
      idx[1] = d; idx[0] = cc; idx[2] = b; idx[3] = a;
    } else { /* k == 2: shape (n1,n0); A.T[c0,d0,col] = G_col[d0,c0] -> i1=d0,i0=c0; row = c0*2+d0 */
      int c0 = (r >> 1) & 1, d0 = r & 1;
      idx[1] = d0; idx[0] = c0;
    }
    keys[r] = 0;
    for (c = 0; c < k; c++)
      if (idx[c]) keys[r] |= (uint8_t)(1u << c);
  }
}

static void draw_spawn_discard(sfo_env* e) {
  /* advance the stream by one resetShip() worth of draws (SRC/game.cpp:133-149) */
  sfo_game tmp;
  sfo_game_init(&tmp, &e->cfg, &e->rng);
}

/* `from`: the libc stream where this env's process stands when its first Game is made, or NULL = srand(seed) advanced by
 * spawn_skip spawns.  (sfo_vec_create hands every env a COPY of one stream it advances once, lane by lane: the same state
 * as seeding each env anew and skipping spawn_skip + i * spawn_stride spawns, in O(n * stride) instead of O(n^2).) */
static int env_init_from(sfo_env* e, const char* gametype, int action_set, int obs_type, unsigned seed, int spawn_skip,
                         const sfo_rng* from) {
  int i;
  memset(e, 0, sizeof(*e));
  if (sfo_preset(gametype, &e->cfg) != 0) return -1;
  e->youturn = (strcmp(gametype, "youturn") == 0 || strcmp(gametype, "test-youturn") == 0); /* ENV:65-66 */
  if (e->youturn) {
    if (action_set == -1 || action_set == 0) { /* ENV:67-70 */
      e->n_actions = 16;
      meshgrid_rows(4, e->action_keys);
    } else if (action_set == 1) { /* ENV:71-78 */
      static const uint8_t t[5] = {0, 1, 2, 4, 8};
      e->n_actions = 5;
      memcpy(e->action_keys, t, 5);
    } else {
      return -2;
    }
  } else {
    if (action_set == -1) { /* ENV:80-81: 16 rows of 4 columns; only columns 0,1 are read (ENV:213-220) */
      e->n_actions = 16;
      meshgrid_rows(4, e->action_keys);
      for (i = 0; i < 16; i++) e->action_keys[i] &= 3;
    } else if (action_set == 0) { /* ENV:82-83 */
      e->n_actions = 4;
      meshgrid_rows(2, e->action_keys);
    } else if (action_set == 1) { /* ENV:84-89 */
      static const uint8_t t[3] = {0, 1, 2};
      e->n_actions = 3;
      memcpy(e->action_keys, t, 3);
    } else {
      return -2;
    }
  }
  e->obs_type = obs_type;
  e->faithful_bugs = 1;
  e->ref_reset_obs = 0;
  e->tickdur = (int)ceil(1. / 30 * 1000); /* ENV:61 */
  e->prev_vlner = 0;                      /* ENV:92 */
  e->pb_width = (int)(450 * .2);          /* ENV:57 */
  e->pb_height = (int)(460 * .2);         /* ENV:58 */
  if (from) {
    e->rng = *from;
  } else {
    sfo_srand(&e->rng, seed);
    for (i = 0; i < spawn_skip; i++) draw_spawn_discard(e);
  }
  sfo_game_init(&e->g, &e->cfg, &e->rng); /* ENV:93 -> reset() -> ENV:164 */
  return 0;
}

int sfo_env_init(sfo_env* e, const char* gametype, int action_set, int obs_type, unsigned seed,
                 int spawn_skip) {
  return env_init_from(e, gametype, action_set, obs_type, seed, spawn_skip, NULL);
}

int sfo_env_obs_dim(const sfo_env* e) {
  if (e->obs_type == SFO_OBS_MONITORS) return 10;
  return 15 + (e->youturn ? 4 : 2);
}

static int count_alive(const sfo_obj* o, int n) {
  int i, c = 0;
  for (i = 0; i < n; i++) c += o[i].alive ? 1 : 0;
  return c;
}

static double py_float_mod(double a, double b) { /* CPython float_rem */
  double m = fmod(a, b);
  if (m) {
    if ((b < 0) != (m < 0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}

static double clip1(double v) { return v < -1 ? -1 : (v > 1 ? 1 : v); }

void sfo_env_features(const sfo_env* e, double* f) { /* ENV:95-157 */
  const sfo_game* g = &e->g;
  int n_missiles = count_alive(g->missiles, SFO_MAX_MISSILES);
  /* SRC/pymodule.cpp:131-134: the `shells` getter walks mMissiles */
  int n_shells = e->faithful_bugs ? n_missiles : count_alive(g->shells, SFO_MAX_SHELLS);
  /* ENV:148 reads vulnerability_timer/_time through "d"-format getters fed ints
   * (SRC/pymodule.cpp:44-45, undefined behaviour); the intended predicate is used. */
  int kill_ready = (g->vlner > 10 && g->fort_vuln_timer < g->cfg.vuln_time) ? 1 : 0;
  int timers[4] = {g->fire_timer, g->thrust_timer, g->left_timer, g->right_timer}; /* SRC/pymodule.cpp:98-105 */
  int nt = e->youturn ? 4 : 2, i;

  if (e->obs_type == SFO_OBS_MONITORS) { /* ENV:96-108 */
    f[0] = n_missiles > 0 ? 0.5 : -0.5;
    f[1] = g->fortress.alive ? 0.5 : -0.5;
    f[2] = g->vlner > 10 ? 0.5 : -0.5;
    f[3] = kill_ready ? 0.5 : -0.5;
    f[4] = g->aim < 3 ? 0.5 : -0.5;
    f[5] = g->aim > 3 ? 0.5 : -0.5;
    f[6] = g->ndist > .75 ? 0.5 : -0.5;
    f[7] = g->ndist > .25 ? 0.5 : -0.5;
    f[8] = g->ndist < -.25 ? 0.5 : -0.5;
    f[9] = g->ndist < -.75 ? 0.5 : -0.5;
    return;
  }
  if (e->obs_type == SFO_OBS_NORMALIZED) { /* ENV:109-133 */
    double max_ticks = floor((double)g->cfg.game_time / e->tickdur); /* ENV:165 */
    f[0] = g->ship.alive ? 1 : 0;
    f[1] = g->ship.pos.x / e->pb_width;
    f[2] = g->ship.pos.y / e->pb_height;
    f[3] = g->ship.vel.x / 10;
    f[4] = g->ship.vel.y / 10;
    f[5] = g->ship.angle / 360;
    f[6] = g->aim / 180;
    f[7] = py_float_mod(g->vdir, 360) / 360;
    f[8] = g->ndist;
    f[9] = g->fortress.alive ? 1 : 0;
    f[10] = g->fortress.angle / 360;
    f[11] = (double)(g->vlner > 10 ? g->vlner : 10) / 10; /* ENV:122 max(), as written */
    f[12] = kill_ready;
    f[13] = (double)n_missiles / SFO_MAX_MISSILES;
    f[14] = (double)n_shells / SFO_MAX_SHELLS;
    for (i = 0; i < nt; i++) f[15 + i] = timers[i] / max_ticks;
    for (i = 0; i < 15 + nt; i++) f[i] = clip1(f[i]);
    return;
  }
  /* ENV:134-157 */
  f[0] = g->ship.alive ? 1 : 0;
  f[1] = g->ship.pos.x;
  f[2] = g->ship.pos.y;
  f[3] = g->ship.vel.x;
  f[4] = g->ship.vel.y;
  f[5] = g->ship.angle;
  f[6] = g->aim;
  f[7] = g->vdir;
  f[8] = g->ndist;
  f[9] = g->fortress.alive ? 1 : 0;
  f[10] = g->fortress.angle;
  f[11] = g->vlner;
  f[12] = kill_ready;
  f[13] = n_missiles;
  f[14] = n_shells;
  for (i = 0; i < nt; i++) f[15 + i] = timers[i];
}

void sfo_env_reset(sfo_env* e, double* obs) { /* ENV:163-178; prev_vlner is NOT cleared */
  sfo_game_init(&e->g, &e->cfg, &e->rng);
  if (e->ref_reset_obs) e->g.vdir = e->g.aim = e->g.fdist = e->g.ndist = 0; /* fresh memory under mExtra, SRC/game.cpp:78 */
  if (obs) sfo_env_features(e, obs);
}

int sfo_env_step(sfo_env* e, int action, double* obs, int* reward, int* done, int* info) { /* ENV:208-253 */
  int r, fort_kill;
  uint8_t keys;
  if (action < 0 || action >= e->n_actions) return -1;
  keys = e->action_keys[action];
  if (keys & 1) sfo_press_key(&e->g, SFO_FIRE_KEY); else sfo_release_key(&e->g, SFO_FIRE_KEY);
  if (keys & 2) sfo_press_key(&e->g, SFO_THRUST_KEY); else sfo_release_key(&e->g, SFO_THRUST_KEY);
  if (e->youturn) {
    if (keys & 4) sfo_press_key(&e->g, SFO_LEFT_KEY); else sfo_release_key(&e->g, SFO_LEFT_KEY);
    if (keys & 8) sfo_press_key(&e->g, SFO_RIGHT_KEY); else sfo_release_key(&e->g, SFO_RIGHT_KEY);
  }
  r = sfo_step_one_tick(&e->g, e->tickdur); /* ENV:231 */
  fort_kill = r > 0;                        /* ENV:233 */
  if (e->cfg.shaped) {                      /* ENV:235-244 */
    int vlner_change = e->g.vlner - e->prev_vlner;
    if (e->g.vlner <= 10 && !fort_kill) r += vlner_change;
    if (r > 1) r = 1;
    if (r < -1) r = -1;
    r = r + 2 * fort_kill;
    e->prev_vlner = e->g.vlner;
  }
  *reward = r;
  *done = sfo_is_game_over(&e->g); /* ENV:246 */
  *info = fort_kill;               /* ENV:253: info is the bare bool */
  if (obs) sfo_env_features(e, obs);
  return 0;
}

/* ------------------------------------------------------------------ */
/* gym_vecenv.SubprocVecEnv worker loop as the trainer relies on it
 * (rl/train.py:60,80): step; if done, the returned observation is the one of
 * the fresh episode, reward/done/info are those of the finished step. */

struct sfo_vec_env {
  int n;
  sfo_env* envs;
};

sfo_vec_env* sfo_vec_create(const char* gametype, int n, int action_set, int obs_type,
                            unsigned seed, int spawn_skip, int spawn_stride) {
  int i;
  sfo_vec_env* v = (sfo_vec_env*)calloc(1, sizeof(*v));
  if (!v) return NULL;
  v->n = n;
  v->envs = (sfo_env*)calloc((size_t)n, sizeof(sfo_env));
  if (!v->envs) { free(v); return NULL; }
  {
    /* one stream, advanced once: lane i starts where srand(seed) stands after spawn_skip + i * spawn_stride spawns */
    sfo_env walker;
    int k;
    if (env_init_from(&walker, gametype, action_set, obs_type, seed, spawn_skip, NULL) != 0) {
      free(v->envs);
      free(v);
      return NULL;
    }
    sfo_srand(&walker.rng, seed); /* (env_init_from drew the walker's own first Game: start over, skip only) */
    for (k = 0; k < spawn_skip; k++) draw_spawn_discard(&walker);
    for (i = 0; i < n; i++) {
      if (env_init_from(&v->envs[i], gametype, action_set, obs_type, seed, 0, &walker.rng) != 0) {
        free(v->envs);
        free(v);
        return NULL;
      }
      v->envs[i].g.rng = &v->envs[i].rng;
      for (k = 0; k < spawn_stride; k++) draw_spawn_discard(&walker);
    }
  }
  return v;
}

void sfo_vec_destroy(sfo_vec_env* v) {
  if (!v) return;
  free(v->envs);
  free(v);
}

int sfo_vec_obs_dim(const sfo_vec_env* v) { return sfo_env_obs_dim(&v->envs[0]); }
sfo_env* sfo_vec_env_at(sfo_vec_env* v, int i) { return &v->envs[i]; }

void sfo_vec_reset(sfo_vec_env* v, double* obs) {
  int i, d = sfo_vec_obs_dim(v);
  for (i = 0; i < v->n; i++) sfo_env_reset(&v->envs[i], obs ? obs + (size_t)i * d : NULL);
}

void sfo_vec_step(sfo_vec_env* v, const int32_t* actions, double* obs, int32_t* reward,
                  uint8_t* done, uint8_t* info) {
  int i, d = sfo_vec_obs_dim(v);
  for (i = 0; i < v->n; i++) {
    int r = 0, dn = 0, inf = 0;
    double* o = obs ? obs + (size_t)i * d : NULL;
    if (sfo_env_step(&v->envs[i], actions[i], o, &r, &dn, &inf) != 0) {
      r = 0; dn = 0; inf = 0; /* out-of-range action: caller validates */
    }
    if (dn) sfo_env_reset(&v->envs[i], o);
    if (reward) reward[i] = r;
    if (done) done[i] = (uint8_t)dn;
    if (info) info[i] = (uint8_t)inf;
  }
}

/* ------------------------------------------------------------------ */
/* Bare engine rollout with uniform random actions from a private LCG: the
 * same loop as sfref_rollout in ref_driver.cpp, used to cross-check long runs
 * against the reference build and as the "port" CPU baseline. */
long sfo_rollout(sfo_env* e, long n_steps, unsigned lcg_seed) {
  static const unsigned char keys5[5] = {0, 1, 2, 4, 8};
  const int n_act = e->youturn ? 5 : 3;
  unsigned long long s = lcg_seed * 2862933555777941757ULL + 3037000493ULL;
  long total = 0, i;
  for (i = 0; i < n_steps; i++) {
    unsigned k;
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    k = keys5[(unsigned)((s >> 33) % (unsigned)n_act)];
    if (k & 1) sfo_press_key(&e->g, SFO_FIRE_KEY); else sfo_release_key(&e->g, SFO_FIRE_KEY);
    if (k & 2) sfo_press_key(&e->g, SFO_THRUST_KEY); else sfo_release_key(&e->g, SFO_THRUST_KEY);
    if (e->youturn) {
      if (k & 4) sfo_press_key(&e->g, SFO_LEFT_KEY); else sfo_release_key(&e->g, SFO_LEFT_KEY);
      if (k & 8) sfo_press_key(&e->g, SFO_RIGHT_KEY); else sfo_release_key(&e->g, SFO_RIGHT_KEY);
    }
    total += sfo_step_one_tick(&e->g, 34);
    if (sfo_is_game_over(&e->g)) sfo_game_init(&e->g, &e->cfg, &e->rng);
  }
  return total;
}

/* small accessors so ctypes users need not mirror struct layouts */
sfo_env* sfo_env_new(const char* gametype, int action_set, int obs_type, unsigned seed, int spawn_skip) {
  sfo_env* e = (sfo_env*)calloc(1, sizeof(sfo_env));
  if (!e) return NULL;
  if (sfo_env_init(e, gametype, action_set, obs_type, seed, spawn_skip) != 0) {
    free(e);
    return NULL;
  }
  e->g.rng = &e->rng;
  return e;
}
void sfo_env_free(sfo_env* e) { free(e); }
sfo_game* sfo_env_game(sfo_env* e) { return &e->g; }
int sfo_env_n_actions(const sfo_env* e) { return e->n_actions; }
int sfo_env_action_keys(const sfo_env* e, int a) { return e->action_keys[a]; }
int sfo_env_prev_vlner(const sfo_env* e) { return e->prev_vlner; }
void sfo_env_set_faithful_bugs(sfo_env* e, int on) { e->faithful_bugs = on; }
void sfo_env_set_ref_reset_obs(sfo_env* e, int on) { e->ref_reset_obs = on; }
void sfo_env_snapshot(const sfo_env* e, sfo_snapshot* s) { sfo_game_snapshot(&e->g, s); }
void sfo_vec_snapshot(sfo_vec_env* v, int i, sfo_snapshot* s) { sfo_game_snapshot(&v->envs[i].g, s); }
int sfo_vec_prev_vlner(sfo_vec_env* v, int i) { return v->envs[i].prev_vlner; }
void sfo_env_hex_points(const sfo_env* e, double* out24) {
  int i;
  for (i = 0; i < 6; i++) {
    out24[2 * i] = e->g.big_hex[i].x;
    out24[2 * i + 1] = e->g.big_hex[i].y;
    out24[12 + 2 * i] = e->g.small_hex[i].x;
    out24[12 + 2 * i + 1] = e->g.small_hex[i].y;
  }
}

/* Replay T wrapper steps in one call (keeps the Python tests fast).  After a
 * done step the env is reset like the vec-env worker does; snaps[t] is the
 * state after step t BEFORE that reset, reset_snaps[k] the state after the k-th
 * reset.  obs (T x obs_dim, may be NULL) is what the worker would return: the
 * fresh episode's observation on done steps.  Returns the number of resets, or
 * -1 - t if action t was out of range. */
int sfo_env_replay(sfo_env* e, const uint8_t* actions, int T, sfo_snapshot* snaps, double* obs,
                   int32_t* reward, uint8_t* done, uint8_t* info, sfo_snapshot* reset_snaps,
                   int max_resets) {
  int t, n_resets = 0, d = sfo_env_obs_dim(e);
  for (t = 0; t < T; t++) {
    int r = 0, dn = 0, inf = 0;
    double* o = obs ? obs + (size_t)t * d : NULL;
    if (sfo_env_step(e, actions[t], o, &r, &dn, &inf) != 0) return -1 - t;
    if (snaps) sfo_game_snapshot(&e->g, &snaps[t]);
    if (reward) reward[t] = r;
    if (done) done[t] = (uint8_t)dn;
    if (info) info[t] = (uint8_t)inf;
    if (dn) {
      sfo_env_reset(e, o);
      if (reset_snaps && n_resets < max_resets) sfo_game_snapshot(&e->g, &reset_snaps[n_resets]);
      n_resets++;
    }
  }
  return n_resets;
}

/* Inverse of sfo_game_snapshot: put an env into a constructed state (edge-case tests that play
 * cannot reach, e.g. all 20 missile slots live).  Key flags/timers/score/stats/projectiles are
 * taken from the snapshot; prev_vlner is set separately. */
void sfo_env_load_snapshot(sfo_env* e, const sfo_snapshot* s, int prev_vlner) {
  sfo_game* g = &e->g;
  int i;
  g->time = s->time;
  g->tick = s->tick;
  g->ship.alive = s->ship_alive;
  g->ship.pos.x = s->ship_x;
  g->ship.pos.y = s->ship_y;
  g->ship.vel.x = s->ship_vx;
  g->ship.vel.y = s->ship_vy;
  g->ship.angle = s->ship_angle;
  g->ship_death_timer = s->ship_death_timer;
  g->fire_timer = s->fire_timer;
  g->thrust_timer = s->thrust_timer;
  g->left_timer = s->left_timer;
  g->right_timer = s->right_timer;
  g->thrust_flag = s->thrust_flag;
  g->fire_flag = s->fire_flag;
  g->left_flag = s->left_flag;
  g->right_flag = s->right_flag;
  g->turn_flag = s->turn_flag;
  g->fortress.alive = s->fort_alive;
  g->fortress.angle = s->fort_angle;
  g->fort_last_angle = s->fort_last_angle;
  g->fort_timer = s->fort_timer;
  g->fort_death_timer = s->fort_death_timer;
  g->fort_vuln_timer = s->fort_vuln_timer;
  g->points = s->points;
  g->raw_points = s->raw_points;
  g->vlner = s->vlner;
  memcpy(&g->stats, s->stats, sizeof(s->stats));
  for (i = 0; i < SFO_MAX_MISSILES; i++) {
    g->missiles[i].alive = s->missile_alive[i];
    g->missiles[i].pos.x = s->missile_x[i];
    g->missiles[i].pos.y = s->missile_y[i];
    g->missiles[i].vel.x = s->missile_vx[i];
    g->missiles[i].vel.y = s->missile_vy[i];
    g->missiles[i].angle = s->missile_angle[i];
    g->missiles[i].radius = g->cfg.missile_radius;
  }
  for (i = 0; i < SFO_MAX_SHELLS; i++) {
    g->shells[i].alive = s->shell_alive[i];
    g->shells[i].pos.x = s->shell_x[i];
    g->shells[i].pos.y = s->shell_y[i];
    g->shells[i].vel.x = s->shell_vx[i];
    g->shells[i].vel.y = s->shell_vy[i];
    g->shells[i].angle = s->shell_angle[i];
    g->shells[i].radius = g->cfg.shell_radius;
  }
  compute_extra(g);
  e->prev_vlner = prev_vlner;
}

void sfo_vec_load_snapshot(sfo_vec_env* v, int i, const sfo_snapshot* s, int prev_vlner) {
  sfo_env_load_snapshot(&v->envs[i], s, prev_vlner);
}
