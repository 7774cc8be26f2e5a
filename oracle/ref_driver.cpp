/*
 * ref_driver.cpp -- thin C driver around the REAL reference engine.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it #includes the reference's
 * headers where they lie under /root/reference and is linked with the
 * reference's own engine translation units (game, vector, object, hexagon,
 * config, configs) by oracle/Makefile.  The output (oracle/_ref/libsfref.so) is
 * git-ignored; no reference source is copied into this repository.
 *
 * The engine headers pull in draw.hh (cairo) only through space-fortress.hh;
 * the Makefile passes -D__SF_CAIRO_H__, draw.hh's own include guard, so the
 * renderer header removes itself.  No stand-in for cairo is written, and
 * draw.cpp / pymodule.cpp (which need cairo / the renderer) are not built.
 *
 * What it gives the tests: the `_spacefortress.Game` surface of
 * SRC/pymodule.cpp:361-370 (press_key, release_key, step_one_tick,
 * is_game_over) as plain C calls, a full-state snapshot in the same flat
 * layout the oracle uses (sfo_snapshot), and a bare random-action rollout loop
 * for the CPU baseline.
 *
 * Each game owns a private libc random state (initstate/setstate), because in
 * the reference every env is its own process with its own rand() stream
 * (SRC/game.cpp:137-148; no srand anywhere => seed 1).
 */
#include <cstdlib>
#include <cstring>
#include <new>

#include "space-fortress.hh"
#include "sf_oracle.h"

namespace {

struct RefGame {
  Config* cfg;
  Game* game;
  void* mem;
  char rng[128];
  char preset[32];
};

Config* make_config(const char* name) { /* SRC/pymodule.cpp:332-343 */
  if (!strcmp(name, "autoturn")) return autoturnConfig();
  if (!strcmp(name, "youturn")) return youturnConfig();
  if (!strcmp(name, "test-youturn")) return testyouturnConfig();
  if (!strcmp(name, "test-autoturn")) return testautoturnConfig();
  return nullptr;
}

void new_game(RefGame* r) {
  if (r->game) {
    r->game->~Game();
    free(r->mem);
  }
  /* The reference leaves Fortress::mVulnerabilityTimer and mExtra
   * uninitialised (SRC/game.cpp:78); a zeroed block makes the run
   * deterministic: the timer starts at 0 + 250. */
  r->mem = calloc(1, sizeof(Game));
  r->game = new (r->mem) Game(r->cfg);
}

}  // namespace

extern "C" {

void* sfref_create(const char* preset, unsigned seed, int spawn_skip) {
  RefGame* r = (RefGame*)calloc(1, sizeof(RefGame));
  r->cfg = make_config(preset);
  if (!r->cfg) {
    free(r);
    return nullptr;
  }
  strncpy(r->preset, preset, sizeof(r->preset) - 1);
  char* prev = initstate(seed, r->rng, sizeof(r->rng));
  for (int i = 0; i < spawn_skip; i++) new_game(r); /* each constructor draws one spawn */
  new_game(r);
  setstate(prev);
  return r;
}

void sfref_destroy(void* h) {
  RefGame* r = (RefGame*)h;
  if (!r) return;
  if (r->game) {
    r->game->~Game();
    free(r->mem);
  }
  delete r->cfg;
  free(r);
}

/* SSF_Env.reset(): a brand-new Game in the same process (ENV:164) */
void sfref_new_game(void* h) {
  RefGame* r = (RefGame*)h;
  char* prev = setstate(r->rng);
  new_game(r);
  setstate(prev);
}

void sfref_press_key(void* h, int sym) { ((RefGame*)h)->game->pressKey((KeySym)sym); }
void sfref_release_key(void* h, int sym) { ((RefGame*)h)->game->releaseKey((KeySym)sym); }

int sfref_step_one_tick(void* h, int ms) {
  RefGame* r = (RefGame*)h;
  char* prev = setstate(r->rng);
  int rew = r->game->stepOneTick(ms);
  setstate(prev);
  return rew;
}

int sfref_is_game_over(void* h) { return ((RefGame*)h)->game->isGameOver() ? 1 : 0; }

// the telemetry vectors behind the `thrust_durations`, `shot_durations`, `shot_intervals_invul`,
// `shot_intervals_vul` getters (SRC/pymodule.cpp:143-181; SRC/game.hh:98-101): which = 0..3
int sfref_durations(void* h, int which, int* out, int cap) {
  Game* g = ((RefGame*)h)->game;
  const std::vector<int>& v = which == 0 ? g->mThrustDurations : which == 1 ? g->mShotDurations
                              : which == 2 ? g->mShotIntervalsInvul : g->mShotIntervalsVul;
  for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
  return (int)v.size();
}

// Game::dumpState() (SRC/game.cpp:519-576), what the `dump` method of the Python type returns
int sfref_dump(void* h, char* buf, int cap) {
  std::string s = ((RefGame*)h)->game->dumpState();
  if (buf && cap > 0) {
    strncpy(buf, s.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
  }
  return (int)s.size();
}

int sfref_sizeof_game(void) { return (int)sizeof(Game); }

/* big hexagon vertices x0,y0..x5,y5 then the small hexagon's (SRC/hexagon.cpp:13-34) */
void sfref_hex_points(void* h, double* out24) {
  const Game* g = ((RefGame*)h)->game;
  for (int i = 0; i < 6; i++) {
    out24[2 * i] = g->mBighex.mPoints[i].mX;
    out24[2 * i + 1] = g->mBighex.mPoints[i].mY;
    out24[12 + 2 * i] = g->mSmallhex.mPoints[i].mX;
    out24[12 + 2 * i + 1] = g->mSmallhex.mPoints[i].mY;
  }
}

void sfref_snapshot(void* h, sfo_snapshot* s) {
  const Game* g = ((RefGame*)h)->game;
  memset(s, 0, sizeof(*s));
  s->time = g->mTime;
  s->tick = g->mTick;
  s->ship_alive = g->mShip.mAlive;
  s->ship_x = g->mShip.mPos.mX;
  s->ship_y = g->mShip.mPos.mY;
  s->ship_vx = g->mShip.mVel.mX;
  s->ship_vy = g->mShip.mVel.mY;
  s->ship_angle = g->mShip.mAngle;
  s->ship_death_timer = g->mShip.mDeathTimer;
  s->fire_timer = g->mShip.mFireTimer;
  s->thrust_timer = g->mShip.mThrustTimer;
  s->left_timer = g->mShip.mLeftTimer;
  s->right_timer = g->mShip.mRightTimer;
  s->thrust_flag = g->mShip.mThrustFlag;
  s->fire_flag = g->mShip.mFireFlag;
  s->left_flag = g->mShip.mLeftFlag;
  s->right_flag = g->mShip.mRightFlag;
  s->turn_flag = g->mShip.mTurnFlag;
  s->fort_alive = g->mFortress.mAlive;
  s->fort_angle = g->mFortress.mAngle;
  s->fort_last_angle = g->mFortress.mLastAngle;
  s->fort_timer = g->mFortress.mTimer;
  s->fort_death_timer = g->mFortress.mDeathTimer;
  s->fort_vuln_timer = g->mFortress.mVulnerabilityTimer;
  s->points = g->mScore.mPoints;
  s->raw_points = g->mScore.mRawPoints;
  s->vlner = g->mScore.mVulnerability;
  s->stats[0] = g->mStats.bigHexDeaths;
  s->stats[1] = g->mStats.smallHexDeaths;
  s->stats[2] = g->mStats.shellDeaths;
  s->stats[3] = g->mStats.shipDeaths;
  s->stats[4] = g->mStats.resets;
  s->stats[5] = g->mStats.destroyedFortresses;
  s->stats[6] = g->mStats.missedShots;
  s->stats[7] = g->mStats.totalShots;
  s->stats[8] = g->mStats.totalThrusts;
  s->stats[9] = g->mStats.totalLefts;
  s->stats[10] = g->mStats.totalRights;
  s->stats[11] = g->mStats.vlnerIncs;
  s->stats[12] = g->mStats.maxVlner;
  s->vdir = g->mExtra.vdir;
  s->fdist = g->mExtra.fdist;
  s->ndist = g->mExtra.ndist;
  s->aim = g->mExtra.aim;
  for (int i = 0; i < MAX_MISSILES; i++) {
    s->missile_alive[i] = g->mMissiles[i].mAlive;
    s->missile_x[i] = g->mMissiles[i].mPos.mX;
    s->missile_y[i] = g->mMissiles[i].mPos.mY;
    s->missile_vx[i] = g->mMissiles[i].mVel.mX;
    s->missile_vy[i] = g->mMissiles[i].mVel.mY;
    s->missile_angle[i] = g->mMissiles[i].mAngle;
  }
  for (int i = 0; i < MAX_SHELLS; i++) {
    s->shell_alive[i] = g->mShells[i].mAlive;
    s->shell_x[i] = g->mShells[i].mPos.mX;
    s->shell_y[i] = g->mShells[i].mPos.mY;
    s->shell_vx[i] = g->mShells[i].mVel.mX;
    s->shell_vy[i] = g->mShells[i].mVel.mY;
    s->shell_angle[i] = g->mShells[i].mAngle;
  }
  s->collisions = (g->mCollisions.bigHex ? 1 : 0) | (g->mCollisions.smallHex ? 2 : 0) |
                  (g->mCollisions.missileFortress ? 4 : 0) | (g->mCollisions.shellShip ? 8 : 0);
}

/* Replay T steps of key bits (bit0 FIRE, 1 THRUST, 2 LEFT, 3 RIGHT; ENV:213-229)
 * in one call; snaps[t] = state after step t, a new Game after game over. */
int sfref_replay(void* h, const unsigned char* keys, int T, sfo_snapshot* snaps, int* eng_reward,
                 unsigned char* done) {
  RefGame* r = (RefGame*)h;
  const bool youturn = !strcmp(r->preset, "youturn") || !strcmp(r->preset, "test-youturn");
  int n_resets = 0;
  char* prev = setstate(r->rng);
  for (int t = 0; t < T; t++) {
    Game* g = r->game;
    unsigned k = keys[t];
    if (k & 1) g->pressKey(FIRE_KEY); else g->releaseKey(FIRE_KEY);
    if (k & 2) g->pressKey(THRUST_KEY); else g->releaseKey(THRUST_KEY);
    if (youturn) {
      if (k & 4) g->pressKey(LEFT_KEY); else g->releaseKey(LEFT_KEY);
      if (k & 8) g->pressKey(RIGHT_KEY); else g->releaseKey(RIGHT_KEY);
    }
    int rew = g->stepOneTick(34);
    if (eng_reward) eng_reward[t] = rew;
    if (snaps) sfref_snapshot(h, &snaps[t]);
    bool over = g->isGameOver();
    if (done) done[t] = over ? 1 : 0;
    if (over) {
      new_game(r);
      n_resets++;
    }
  }
  setstate(prev);
  return n_resets;
}

/* Bare engine rollout for the CPU baseline: the key traffic of SSF_Env.step
 * (ENV:213-231) with uniform random actions from a private LCG, a new Game at
 * game over (ENV:164).  Returns the sum of engine rewards so the loop cannot be
 * optimised away.  youturn: 5 actions x 4 keys; autoturn: 3 actions x 2 keys. */
long sfref_rollout(void* h, long n_steps, unsigned lcg_seed) {
  RefGame* r = (RefGame*)h;
  const bool youturn = !strcmp(r->preset, "youturn") || !strcmp(r->preset, "test-youturn");
  const int n_act = youturn ? 5 : 3;
  static const unsigned char keys5[5] = {0, 1, 2, 4, 8};
  unsigned long long s = lcg_seed * 2862933555777941757ULL + 3037000493ULL;
  long total = 0;
  char* prev = setstate(r->rng);
  for (long i = 0; i < n_steps; i++) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    unsigned k = keys5[(unsigned)((s >> 33) % (unsigned)n_act)];
    Game* g = r->game;
    if (k & 1) g->pressKey(FIRE_KEY); else g->releaseKey(FIRE_KEY);
    if (k & 2) g->pressKey(THRUST_KEY); else g->releaseKey(THRUST_KEY);
    if (youturn) {
      if (k & 4) g->pressKey(LEFT_KEY); else g->releaseKey(LEFT_KEY);
      if (k & 8) g->pressKey(RIGHT_KEY); else g->releaseKey(RIGHT_KEY);
    }
    total += g->stepOneTick(34);
    if (g->isGameOver()) new_game(r);
  }
  setstate(prev);
  return total;
}

}  // extern "C"

/* Write a snapshot INTO the reference game (its members are public): every field sfref_snapshot reads.
 * Lets the fixtures place the reference's own objects anywhere (a ship at every heading, crowded pools). */
extern "C" void sfref_load_snapshot(void* h, const sfo_snapshot* s) {
  Game* g = ((RefGame*)h)->game;
  g->mTime = s->time;
  g->mTick = s->tick;
  g->mShip.mAlive = s->ship_alive;
  g->mShip.mPos.mX = s->ship_x;
  g->mShip.mPos.mY = s->ship_y;
  g->mShip.mVel.mX = s->ship_vx;
  g->mShip.mVel.mY = s->ship_vy;
  g->mShip.mAngle = s->ship_angle;
  g->mShip.mDeathTimer = s->ship_death_timer;
  g->mFortress.mAlive = s->fort_alive;
  g->mFortress.mAngle = s->fort_angle;
  g->mFortress.mLastAngle = s->fort_last_angle;
  g->mFortress.mTimer = s->fort_timer;
  g->mFortress.mDeathTimer = s->fort_death_timer;
  g->mFortress.mVulnerabilityTimer = s->fort_vuln_timer;
  g->mScore.mPoints = s->points;
  g->mScore.mRawPoints = s->raw_points;
  g->mScore.mVulnerability = s->vlner;
  for (int i = 0; i < MAX_MISSILES; i++) {
    g->mMissiles[i].mAlive = s->missile_alive[i];
    g->mMissiles[i].mPos.mX = s->missile_x[i];
    g->mMissiles[i].mPos.mY = s->missile_y[i];
    g->mMissiles[i].mVel.mX = s->missile_vx[i];
    g->mMissiles[i].mVel.mY = s->missile_vy[i];
    g->mMissiles[i].mAngle = s->missile_angle[i];
  }
  for (int i = 0; i < MAX_SHELLS; i++) {
    g->mShells[i].mAlive = s->shell_alive[i];
    g->mShells[i].mPos.mX = s->shell_x[i];
    g->mShells[i].mPos.mY = s->shell_y[i];
    g->mShells[i].mVel.mX = s->shell_vx[i];
    g->mShells[i].mVel.mY = s->shell_vy[i];
    g->mShells[i].mAngle = s->shell_angle[i];
  }
}

#ifdef SFREF_WITH_DRAW
/* Only in oracle/_ref/libsfrefdraw.so (`make refdraw`): the reference's REAL renderer -- draw.cpp and wireframe.cpp
 * compiled where they lie and linked with the image's cairo 1.16 (/opt/conda).  What `Game.draw` + `pb_pixels` do
 * (SRC/pymodule.cpp:242-254, :351): newPixelBuffer(width, height, viewport, line width, grayscale), initWireframes()
 * (:485), drawGameStateScaled (SRC/draw.cpp:256-270); out = byte 0 of every RGB24 pixel, which in grayscale mode is
 * what cv2.COLOR_RGBA2GRAY leaves of it (ENV:205: R = G = B).  `out` holds width * height bytes, row-major. */
#include <cairo/cairo.h>
extern "C" int sfref_draw_geom(void* h, int width, int height, int vp_x, int vp_y, int vp_w, int vp_h,
                               double line_width, int grayscale, int channel, unsigned char* out) {
  initWireframes();
  PixelBuffer* pb = newPixelBuffer(width, height, vp_x, vp_y, vp_w, vp_h, line_width, grayscale != 0);
  if (!pb) return -1;
  drawGameStateScaled(((RefGame*)h)->game, pb);
  cairo_surface_flush(pb->surface);
  const unsigned char* raw = (const unsigned char*)pb->raw;
  for (int y = 0; y < pb->height; y++)
    for (int x = 0; x < pb->width; x++) out[y * pb->width + x] = raw[y * pb->stride + 4 * x + channel];
  freePixelBuffer(pb);
  return 0;
}

/* SSF_Env's default: SSF_Env(scale=.2, viewport=(130,80,450,460), ls=3) -> Game(..., 3.0, True, 90, 92, viewport) (ENV:57-60) */
extern "C" int sfref_draw(void* h, unsigned char* out) {
  return sfref_draw_geom(h, 90, 92, 130, 80, 450, 460, 3.0, 1, 0, out);
}

extern "C" const char* sfref_cairo_version(void) { return cairo_version_string(); }
#endif
