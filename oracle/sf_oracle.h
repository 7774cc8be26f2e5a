/*
 * sf_oracle.h -- CPU restatement of the Space Fortress env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker (or as the timed CPU baseline),
 * never as the thing shipped.  The product path (spacefortress_amd/, libsfmi.so)
 * neither links nor calls it.
 *
 * Parity status: PINNED.  The restatement is checked field by field against the
 * reference C++ engine itself, built from /root/reference by oracle/Makefile
 * into oracle/_ref/ (see oracle/ref_driver.cpp), and against the golden vectors
 * under tests/golden/ that were generated from that build.
 *
 * Citations "SRC/..." are relative to /root/reference/python/spacefortress/src,
 * "ENV:" is /root/reference/python/spacefortress.gym/spacefortress/gym/envs/ssf_env.py.
 *
 * Plain C99, scalar, one environment at a time -- deliberately simple.
 */
#ifndef SF_ORACLE_H
#define SF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFO_MAX_MISSILES 20 /* SRC/game.hh:3 */
#define SFO_MAX_SHELLS 20   /* SRC/game.hh:4 */
#define SFO_MAX_KEY_EVENTS 64

/* SRC/game.hh:15-17 */
enum { SFO_NO_KEY = 0, SFO_FIRE_KEY = 1, SFO_THRUST_KEY = 2, SFO_LEFT_KEY = 3, SFO_RIGHT_KEY = 4 };
/* SRC/game.hh:13 */
enum { SFO_NO_TURN = 0, SFO_TURN_LEFT = 1, SFO_TURN_RIGHT = 2 };

/* The preset values the path reads (SRC/configs.cpp:3-89); the string-keyed
 * map of SRC/config.cpp is not reproduced. */
typedef struct {
  int width, height, game_time;
  int destroy_fortress, ship_death_penalty;
  double missile_penalty;
  int miss_penalty;
  int shell_speed, shell_radius;
  int missile_speed, missile_radius;
  int auto_turn;
  int sector_size, lock_time, vuln_time, vuln_threshold, fortress_radius;
  int big_hex, small_hex;
  int explode_duration;
  double start_vx, start_vy;
  int ship_radius;
  double ship_accel;
  int turn_speed;
  int shaped; /* 1 for "autoturn"/"youturn": ENV:235 applies reward shaping */
} sfo_config;

/* glibc random_r TYPE_3 state (r[i] = r[i-3] + r[i-31]); one per environment,
 * because every reference env lives in its own process with its own libc state. */
typedef struct {
  int32_t r[31];
  int f, b;
} sfo_rng;

typedef struct { double x, y; } sfo_vec;

/* SRC/object.hh:3-14 */
typedef struct {
  sfo_vec pos, vel;
  double angle;
  int radius;
  int alive;
} sfo_obj;

/* SRC/game.hh:29-43 */
typedef struct {
  int big_hex_deaths, small_hex_deaths, shell_deaths, ship_deaths, resets,
      destroyed_fortresses, missed_shots, total_shots, total_thrusts,
      total_lefts, total_rights, vlner_incs, max_vlner;
} sfo_stats;

typedef struct { int sym, state; } sfo_key;

/* SRC/game.hh:84-107, minus strings/telemetry vectors/renderer */
typedef struct {
  sfo_config cfg;
  sfo_rng* rng; /* borrowed: the "process" this game lives in */
  sfo_key events[SFO_MAX_KEY_EVENTS];
  int n_events, events_processed;
  /* ship: SRC/game.hh:58-73 */
  sfo_obj ship;
  int ship_death_timer, fire_timer, thrust_timer, left_timer, right_timer;
  int thrust_flag, fire_flag, left_flag, right_flag, turn_flag;
  /* fortress: SRC/game.hh:75-82 */
  sfo_obj fortress;
  int fort_death_timer, fort_vuln_timer, fort_timer;
  double fort_last_angle;
  sfo_obj missiles[SFO_MAX_MISSILES];
  sfo_obj shells[SFO_MAX_SHELLS];
  sfo_vec big_hex[6], small_hex[6];
  int tick, time;
  int col_big_hex, col_small_hex, col_missile_fortress, col_shell_ship;
  sfo_stats stats;
  double vdir, fdist, ndist, aim;
  float points, raw_points;
  int vlner;
  float reward;
} sfo_game;

/* --- RNG (glibc srandom_r / random_r TYPE_3) --- */
void sfo_srand(sfo_rng* s, unsigned seed);
int sfo_rand(sfo_rng* s);

/* --- presets --- */
/* returns 0 on success, -1 for an unknown name (SRC/pymodule.cpp:332-343) */
int sfo_preset(const char* name, sfo_config* out);

/* --- engine (SRC/game.cpp) --- */
void sfo_game_init(sfo_game* g, const sfo_config* cfg, sfo_rng* rng);
void sfo_press_key(sfo_game* g, int sym);
void sfo_release_key(sfo_game* g, int sym);
int sfo_step_one_tick(sfo_game* g, int ms);
int sfo_is_game_over(const sfo_game* g);

/* --- gym wrapper restatement (ENV) --- */
#define SFO_OBS_FEATURES 0
#define SFO_OBS_NORMALIZED 1
#define SFO_OBS_MONITORS 2

typedef struct {
  sfo_game g;
  sfo_rng rng;
  sfo_config cfg;
  int youturn;          /* 4 keys per action instead of 2 (ENV:65-66,221) */
  int n_actions;
  uint8_t action_keys[16]; /* bit0 FIRE bit1 THRUST bit2 LEFT bit3 RIGHT (ENV:64-89) */
  int obs_type;
  int faithful_bugs;    /* 1: n_shells reports the missile count (SRC/pymodule.cpp:131-134) */
  int tickdur;          /* ENV:61 -> 34 */
  int prev_vlner;       /* ENV:92, survives reset() */
  int pb_width, pb_height; /* ENV:57-58 -> 90, 92 (normalized-features divisor) */
  int ref_reset_obs;    /* 1: a new Game's mExtra reads 0, 0, 0 until its first tick -- what the reference's reset() returns on
                           fresh memory (SRC/game.cpp:78 leaves it unwritten; ENV:163-178); 0: computeExtra(spawn state) */
} sfo_env;

/* action_set follows ENV:67-89 (1 = reduced set, 0/-1 = all key combinations) */
int sfo_env_init(sfo_env* e, const char* gametype, int action_set, int obs_type, unsigned seed,
                 int spawn_skip);
int sfo_env_obs_dim(const sfo_env* e);
void sfo_env_reset(sfo_env* e, double* obs);
/* returns 0, or -1 if the action is out of range (reference: IndexError) */
int sfo_env_step(sfo_env* e, int action, double* obs, int* reward, int* done, int* info);
void sfo_env_features(const sfo_env* e, double* obs);

/* --- SubprocVecEnv-shaped batch: step every env, auto-reset on done --- */
typedef struct sfo_vec_env sfo_vec_env;
sfo_vec_env* sfo_vec_create(const char* gametype, int n, int action_set, int obs_type,
                            unsigned seed, int spawn_skip, int spawn_stride);
void sfo_vec_destroy(sfo_vec_env* v);
int sfo_vec_obs_dim(const sfo_vec_env* v);
void sfo_vec_reset(sfo_vec_env* v, double* obs);
void sfo_vec_step(sfo_vec_env* v, const int32_t* actions, double* obs, int32_t* reward,
                  uint8_t* done, uint8_t* info);
sfo_env* sfo_vec_env_at(sfo_vec_env* v, int i);

/* Flat per-env snapshot used by the parity tests (same field order for the
 * reference driver in oracle/ref_driver.cpp). */
typedef struct {
  int32_t time, tick;
  int32_t ship_alive;
  double ship_x, ship_y, ship_vx, ship_vy, ship_angle;
  int32_t ship_death_timer, fire_timer, thrust_timer, left_timer, right_timer;
  int32_t thrust_flag, fire_flag, left_flag, right_flag, turn_flag;
  int32_t fort_alive;
  double fort_angle, fort_last_angle;
  int32_t fort_timer, fort_death_timer, fort_vuln_timer;
  float points, raw_points;
  int32_t vlner;
  int32_t stats[13];
  double vdir, fdist, ndist, aim;
  int32_t missile_alive[SFO_MAX_MISSILES];
  double missile_x[SFO_MAX_MISSILES], missile_y[SFO_MAX_MISSILES];
  double missile_vx[SFO_MAX_MISSILES], missile_vy[SFO_MAX_MISSILES];
  double missile_angle[SFO_MAX_MISSILES];
  int32_t shell_alive[SFO_MAX_SHELLS];
  double shell_x[SFO_MAX_SHELLS], shell_y[SFO_MAX_SHELLS];
  double shell_vx[SFO_MAX_SHELLS], shell_vy[SFO_MAX_SHELLS];
  double shell_angle[SFO_MAX_SHELLS];
  int32_t collisions; /* bit0 bigHex bit1 smallHex bit2 missileFortress bit3 shellShip */
} sfo_snapshot;

void sfo_game_snapshot(const sfo_game* g, sfo_snapshot* s);
int sfo_snapshot_size(void);

/* bare engine loop with LCG random actions (matches sfref_rollout) */
long sfo_rollout(sfo_env* e, long n_steps, unsigned lcg_seed);

/* heap helpers / accessors for ctypes */
sfo_env* sfo_env_new(const char* gametype, int action_set, int obs_type, unsigned seed, int spawn_skip);
void sfo_env_free(sfo_env* e);
sfo_game* sfo_env_game(sfo_env* e);
int sfo_env_n_actions(const sfo_env* e);
int sfo_env_action_keys(const sfo_env* e, int a);
int sfo_env_prev_vlner(const sfo_env* e);
void sfo_env_set_faithful_bugs(sfo_env* e, int on);
void sfo_env_set_ref_reset_obs(sfo_env* e, int on); /* takes effect at the next reset */
void sfo_env_snapshot(const sfo_env* e, sfo_snapshot* s);
void sfo_vec_snapshot(sfo_vec_env* v, int i, sfo_snapshot* s);
int sfo_vec_prev_vlner(sfo_vec_env* v, int i);
void sfo_env_hex_points(const sfo_env* e, double* out24);
void sfo_env_load_snapshot(sfo_env* e, const sfo_snapshot* s, int prev_vlner);
void sfo_vec_load_snapshot(sfo_vec_env* v, int i, const sfo_snapshot* s, int prev_vlner);
int sfo_env_replay(sfo_env* e, const uint8_t* actions, int T, sfo_snapshot* snaps, double* obs,
                   int32_t* reward, uint8_t* done, uint8_t* info, sfo_snapshot* reset_snaps,
                   int max_resets);

#ifdef __cplusplus
}
#endif
#endif
