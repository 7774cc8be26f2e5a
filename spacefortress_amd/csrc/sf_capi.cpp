// sf_capi.cpp -- the C ABI of include/sfmi.h: batch lifecycle, step/reset launches, state access.
// Host code only; the kernels are in sf_kernels.hip.  There is no CPU implementation of the
// path in this library: every entry point that computes needs a HIP device.
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sf_drawrec.h"
#include "sf_internal.h"
#include "sf_raster.h"

struct sf_batch {
  SfKernelArgs args;
  sf_preset preset;
  bool autoturn;
  int device;
  int n_envs;
  int act_count;
  int spawn_skip, spawn_stride;
  unsigned char* d_state;
  size_t state_bytes;
  double* d_consts;
  int16_t* d_spawn;
  unsigned long long* d_acc;
  unsigned char* d_scratch;  // linear staging for sf_get_field / sf_set_field (largest field)
  int obs_mode;              // the caller's SF_OBS_*; args.obs_type is SF_OBS_NONE for the image modes
  uint32_t* d_bg;            // image observation: static background, 92*90 bytes
  uint32_t* d_bg84;          // ... resampled to 84x84
  double* d_arcs;            // the explosion's arc constants (sf_arc_table), then nothing
  unsigned char* d_falpha;   // the live fortress's coverage at its 36 headings (sf_image_fort_alpha)
  uint32_t* d_tabs;          // INTER_AREA taps (sf_raster.h)
  unsigned char* d_xcache;   // explosion cache, SF_XC_BYTES per env; allocated by the first render
  // The missile fields as the reference has them (per env and slot), kept only while a caller looks at or edits them
  // through sf_get_field / sf_set_field: [SF_NSLOT][n_envs] (x, y) pairs and int32 headings.  The kernels keep a tile's
  // live missiles as one dense pool (sf_layout.h); `mslots_dirty` = the view was edited and the pools are rebuilt from it
  // (and from the alive masks) before the next launch that reads the state.
  unsigned char* d_ms_pos;
  int32_t* d_ms_ang;
  bool mslots_dirty;
  uint32_t* d_actrec;        // sf_step_sampled: one (tick, key0, key1, first lane) record per tile
  // The envs' draw records (sf_drawrec.h), SF_DR_BYTES per lane: what the frame kernel reads.  Image batches have them from
  // sf_create on and their step launches keep them current (args.draw); any other change of the state -- sf_reset,
  // sf_set_field, the missile view -- clears `draw_current`, and the next frame rebuilds them from the state first
  // (sf_drawrec_kernel).  Batches that step with a symbolic observation get the buffer with their first frame and rebuild
  // before every frame.
  unsigned char* d_draw;
  bool draw_current;
  // sf_set_image_geometry: a geometry other than the default one (sf_render_generic.hip); g_w == 0: the default
  int g_w, g_h;
  double g_sx, g_sy, g_vx, g_vy, g_lw;  // scale_x = g_w / vp_w, scale_y = g_h / vp_h: NOT the caller's scale when vp_w * scale is not whole (SRC/draw.cpp:70-71)
  uint8_t* d_gbg;            // g_w * g_h bytes: the hexagons
  uint32_t* d_gtabs;         // the INTER_AREA taps g_w -> 84, g_h -> 84
  bool render_ready;         // the render caches and pictures exist (or were declined: SFMI_NO_EXPLOSION_CACHE)
  // the score text's glyph atlas of the CURRENT geometry (sf_glyphs.h; gw == 0: the seven-segment fallback), its device copy,
  // and the atlas the default geometry's static backgrounds and cached pictures were made with
  SfGlyphAtlas h_glyphs, baked_glyphs;
  SfGlyphAtlas* d_glyphs;
};

namespace {

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      sf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SF_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

// keep the caller's (PyTorch's) current device untouched
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

// rebuild the tiles' missile pools from the edited slot view, in stream order ahead of whatever reads the state next
int flush_missile_view(sf_batch* b, hipStream_t stream) {
  if (!b->mslots_dirty) return SF_OK;
  HIP_TRY(sf_launch_slots_to_mpool(b->d_state, b->args.lanes, b->n_envs, b->d_ms_pos, b->d_ms_ang, stream));
  b->mslots_dirty = false;
  b->draw_current = false;
  return SF_OK;
}
#define SF_FLUSH_VIEW(b, stream)                                   \
  do {                                                             \
    int rc_ = flush_missile_view((b), (hipStream_t)(stream));      \
    if (rc_ != SF_OK) return rc_;                                  \
  } while (0)

const unsigned long long kAccInit[SF_ACC_WORDS] = {
    0, 0, 0, 0, 0, 0, (unsigned long long)LLONG_MAX, (unsigned long long)LLONG_MIN, 0, 0, 0};
static_assert(SF_ACC_BAD_ACTION == SF_EPISODE_STATS_LEN && SF_ACC_OVERFLOW == SF_EPISODE_STATS_LEN + 1, "acc layout");

// the sampler records of sf_step_sampled (sf_layout.h: SF_ACT_SAMPLED): per tile (tick, key0, key1, first lane of the job)
int write_action_records(sf_batch* b, uint64_t seed, uint32_t first_lane, hipStream_t stream) {
  const size_t tiles = (size_t)(b->args.lanes / 64);
  std::vector<uint32_t> rec(4 * tiles);
  for (size_t t = 0; t < tiles; t++) {
    rec[4 * t + 0] = 0u;
    rec[4 * t + 1] = (uint32_t)seed;
    rec[4 * t + 2] = (uint32_t)(seed >> 32);
    rec[4 * t + 3] = first_lane + (uint32_t)(64 * t);
  }
  HIP_TRY(hipMemcpyAsync(b->d_actrec, rec.data(), rec.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipStreamSynchronize(stream));  // `rec` is pageable host memory that dies with this call
  return SF_OK;
}

bool is_pow2(long v) { return v > 0 && (v & (v - 1)) == 0; }

// What a batch needs to draw frames, made once: the per-env explosion cache followed by the 36 fortress pictures, the
// destroyed fortress's explosion (in the layout of an env's cache entry) and the score / bar pictures -- all drawn here, on
// `stream`, by the frame kernel's own code; the 36 x 4 backgrounds with the fortress in them; the envs' draw records.
// Image batches call this from sf_create and wait for it (every later launch, on whatever stream, finds the pictures
// finished: a first frame inside a HIP-graph capture allocates nothing); other batches with their first frame.
// SFMI_NO_EXPLOSION_CACHE (diagnostics, tools/render_soak.py): no caches, no pictures -- every frame drawn in place.
int ensure_render_resources(sf_batch* b, hipStream_t stream) {
  if (b->render_ready) return SF_OK;
  if (!b->d_draw) {
    const size_t bytes = (size_t)b->args.lanes * SF_DR_BYTES;
    HIP_TRY(hipMalloc((void**)&b->d_draw, bytes));
    HIP_TRY(hipMemsetAsync(b->d_draw, 0, bytes, stream));
  }
  if (!b->d_xcache && !getenv("SFMI_NO_EXPLOSION_CACHE")) {
    const size_t bytes = (size_t)b->n_envs * SF_XC_BYTES, tail = 36 * SF_FP_BYTES + SF_XC_BYTES + SF_HUD_BYTES;
    HIP_TRY(hipMalloc((void**)&b->d_xcache, bytes + tail));
    HIP_TRY(hipMemsetAsync(b->d_xcache, 0, bytes + tail, stream));
    HIP_TRY(sf_launch_hud_pictures(b->d_bg, b->d_bg84, b->d_tabs, b->d_xcache + bytes + 36 * SF_FP_BYTES + SF_XC_BYTES, b->d_glyphs, stream));
    HIP_TRY(sf_launch_fort_patches(b->d_bg, b->d_bg84, b->d_tabs, b->d_xcache + bytes, b->d_arcs, b->d_falpha, stream));
  }
  HIP_TRY(hipStreamSynchronize(stream));
  b->render_ready = true;
  b->draw_current = false;
  return SF_OK;
}

// What the default geometry's frame kernel keeps of the score text, (re)made from b->h_glyphs: the four static backgrounds
// (variant bit 0 = the score 0000000 baked in) with their 84x84 images and -- where they exist already -- the explosion
// cache, the score / bar pictures and the backgrounds with the fortress in them.  Synchronous.
int bake_default_glyphs(sf_batch* b) {
  std::vector<uint8_t> bg(4 * SF_BG_STRIDE, 0), bg84(4 * SF_OUT * SF_OUT);
  for (int v = 0; v < 4; v++) {  // static background variants: sf_raster.h
    int rc = sf_image_static_glyphs(v, &b->h_glyphs, bg.data() + v * SF_BG_STRIDE);
    if (rc == SF_OK) rc = sf_resize_area_u8(bg.data() + v * SF_BG_STRIDE, SF_IMG_W, SF_IMG_H, bg84.data() + v * SF_OUT * SF_OUT, SF_OUT, SF_OUT);
    if (rc != SF_OK) return rc;
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(b->d_bg84, bg84.data(), bg84.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_bg, bg.data(), bg.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_glyphs, &b->h_glyphs, sizeof(SfGlyphAtlas), hipMemcpyHostToDevice));
  b->baked_glyphs = b->h_glyphs;
  if (b->render_ready && b->d_xcache) {
    const size_t bytes = (size_t)b->n_envs * SF_XC_BYTES, tail = 36 * SF_FP_BYTES + SF_XC_BYTES + SF_HUD_BYTES;
    HIP_TRY(hipMemset(b->d_xcache, 0, bytes + tail));
    HIP_TRY(sf_launch_hud_pictures(b->d_bg, b->d_bg84, b->d_tabs, b->d_xcache + bytes + 36 * SF_FP_BYTES + SF_XC_BYTES, b->d_glyphs, nullptr));
    HIP_TRY(sf_launch_fort_patches(b->d_bg, b->d_bg84, b->d_tabs, b->d_xcache + bytes, b->d_arcs, b->d_falpha, nullptr));
    HIP_TRY(hipDeviceSynchronize());
  }
  b->draw_current = false;
  return SF_OK;
}

}  // namespace

extern "C" int sf_create(const sf_create_params* p, sf_batch** out) {
  if (!p || !out || !p->gametype) {
    sf_set_error("sf_create: null argument");
    return SF_ERR_ARG;
  }
  *out = nullptr;
  sf_preset preset;
  int rc = sf_preset_get(p->gametype, &preset);
  if (rc != SF_OK) return rc;
  if (p->n_envs <= 0) {
    sf_set_error("sf_create: n_envs must be positive (got %d)", p->n_envs);
    return SF_ERR_ARG;
  }
  if (p->obs_type < SF_OBS_FEATURES || p->obs_type > SF_OBS_IMAGE_RAW) {
    // ENV:51 assert obs_type in ('image', 'features', 'normalized-features', 'monitors')
    sf_set_error("sf_create: unsupported obs_type %d", p->obs_type);
    return SF_ERR_ARG;
  }
  uint8_t keys[16];
  int n_actions = sf_action_table(p->gametype, p->action_set, keys);
  if (n_actions < 0) return n_actions;
  if (p->spawn_skip < 0 || p->spawn_stride < 0) {
    sf_set_error("sf_create: spawn_skip / spawn_stride must be >= 0");
    return SF_ERR_ARG;
  }
  // Lane i starts at entry spawn_skip + spawn_stride * i of the accepted-spawn sequence and must be able to walk on
  // from there as the libc stream does (SRC/game.cpp:133-149), not wrap into somebody else's stretch: the default
  // table reaches SF_SPAWN_MARGIN entries (about 370 episodes of respawns) past the LAST lane's start.
  const long last_start = (long)p->spawn_skip + (long)p->spawn_stride * ((long)p->n_envs - 1);
  long spawn_len = p->spawn_table_len;
  if (spawn_len == 0) {
    spawn_len = 65536;
    while (spawn_len < last_start + SF_SPAWN_MARGIN && spawn_len < (1l << 24)) spawn_len <<= 1;
  }
  if (!is_pow2(spawn_len) || spawn_len > (1l << 24)) {
    sf_set_error("sf_create: spawn_table_len must be a power of two <= 2^24 (got %ld)", spawn_len);
    return SF_ERR_ARG;
  }
  if (last_start >= spawn_len) {
    sf_set_error("sf_create: lane %d would start at entry %ld of a %ld-entry spawn table (spawn_skip %d, spawn_stride %d): "
                 "pass a larger spawn_table_len (<= 2^24) or smaller offsets", p->n_envs - 1, last_start, spawn_len,
                 p->spawn_skip, p->spawn_stride);
    return SF_ERR_ARG;
  }

  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0) {
    sf_set_error("sf_create: no HIP device available (%s); libsfmi has no CPU path",
                 e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    return SF_ERR_NO_DEVICE;
  }
  if (p->device_id < 0 || p->device_id >= n_dev) {
    sf_set_error("sf_create: device_id %d out of range (%d devices)", p->device_id, n_dev);
    return SF_ERR_ARG;
  }
  DeviceGuard guard(p->device_id);

  sf_batch* b = new sf_batch();
  memset(b, 0, sizeof(*b));
  b->preset = preset;
  b->autoturn = preset.auto_turn != 0;
  b->device = p->device_id;
  b->n_envs = p->n_envs;
  b->act_count = n_actions;
  b->spawn_skip = p->spawn_skip;
  b->spawn_stride = p->spawn_stride;

  const long lanes = ((long)p->n_envs + 255) / 256 * 256;
  b->state_bytes = (size_t)sfl::kBytesPerLane * lanes;

  // host tables
  std::vector<double> consts(SF_CONST_DOUBLES);
  sf_host_fill_consts(preset, consts.data());
  std::vector<int16_t> spawn(4 * (size_t)spawn_len);
  rc = sf_spawn_table(p->seed, (int)spawn_len, spawn.data());
  if (rc != SF_OK) {
    delete b;
    return rc;
  }

#define HIP_TRY_FREE(expr)                                                             \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      sf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      sf_destroy(b);                                                                   \
      return SF_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

  HIP_TRY_FREE(hipMalloc((void**)&b->d_state, b->state_bytes));
  HIP_TRY_FREE(hipMemset(b->d_state, 0, b->state_bytes));  // dead projectile slots are read (and ignored): keep them defined
  HIP_TRY_FREE(hipMalloc((void**)&b->d_consts, consts.size() * sizeof(double)));
  HIP_TRY_FREE(hipMalloc((void**)&b->d_spawn, spawn.size() * sizeof(int16_t)));
  HIP_TRY_FREE(hipMalloc((void**)&b->d_acc, sizeof(kAccInit)));
  HIP_TRY_FREE(hipMalloc((void**)&b->d_scratch, (size_t)p->n_envs * SF_NSLOT * 16));
  HIP_TRY_FREE(hipMemcpy(b->d_consts, consts.data(), consts.size() * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY_FREE(hipMemcpy(b->d_spawn, spawn.data(), spawn.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  HIP_TRY_FREE(hipMemcpy(b->d_acc, kAccInit, sizeof(kAccInit), hipMemcpyHostToDevice));
  HIP_TRY_FREE(hipMalloc((void**)&b->d_actrec, (size_t)(lanes / 64) * 16));
  {
    // image observation tables (sf_image.cpp): 11 KB, built for every batch so that sf_render works
    // whatever obs_type the batch steps with (the reference's render(), ENV:190-193)
    static_assert(SF_IMG_W == SF_IMAGE_W && SF_IMG_H == SF_IMAGE_H && SF_OUT == SF_IMAGE_OUT, "sfmi.h vs sf_raster.h");
    static_assert(SF_IMG_W == (int)sfc::pb_width && SF_IMG_H == (int)sfc::pb_height, "ENV:57-58");
    // device layout of the INTER_AREA tables: sf_raster.h
    std::vector<uint32_t> tabs(SF_TAB_WORDS, 0u);
    {
      const int ssize[2] = {SF_IMG_W, SF_IMG_H};
      bool ok = true;
      for (int ax = 0; ax < 2; ax++) {
        int32_t first[SF_OUT], count[SF_OUT];
        float alpha[SF_OUT * 4];
        sf_resize_area_tab(ssize[ax], SF_OUT, first, count, alpha);
        for (int i = 0; i < SF_OUT; i++) {
          // what the kernel's sparse resampling assumes: two column taps; <= three row taps; and the
          // windows where its dirty-box arithmetic expects them (first in [i*s - 1, i*s], s = 15/14, 23/21)
          const double lo = (double)i * ssize[ax] / SF_OUT;
          ok = ok && count[i] <= (ax == 0 ? 2 : 3) && first[i] >= (int)floor(lo) - 1 && first[i] <= (int)floor(lo) &&
               first[i] + count[i] <= ssize[ax];
          uint32_t* t = &tabs[4 * (ax * SF_OUT + i)];
          t[0] = (uint32_t)first[i];
          memcpy(t + 1, alpha + 4 * i, 3 * sizeof(float));
          // ... and that the taps repeat exactly (90 / 84 = 15 / 14, 92 / 84 = 23 / 21): the frame kernel keeps ONE period in
          // LDS (sf_render.hip: resample_into)
          const int P = ax == 0 ? 14 : 21, Q = ax == 0 ? 15 : 23;
          if (i >= P)
            ok = ok && first[i] == first[i - P] + Q && count[i] == count[i - P] &&
                 memcmp(alpha + 4 * i, alpha + 4 * (i - P), 3 * sizeof(float)) == 0;
          // ... and what its exact dirty box relies on (sf_render.hip: out_box): a source cell s is read with a weight
          // that is not zero only by the destinations floor(N s / D) ... ceil(N (s + 1) / D) - 1, N / D = 14/15, 21/23
          const int N = ax == 0 ? 14 : 21, D = ax == 0 ? 15 : 23;
          for (int j = 0; j < count[i]; j++) {
            const int sc = first[i] + j;
            if (alpha[4 * i + j] != 0.0f) ok = ok && i >= (sc * N) / D && i < ((sc + 1) * N + D - 1) / D;
          }
        }
      }
      if (!ok) {
        sf_set_error("sf_create: INTER_AREA table does not have the structure the render kernel assumes");
        sf_destroy(b);
        return SF_ERR_ARG;
      }
    }
    // (room for the backgrounds with the fortress in them, filled by the first frame: sf_launch_fort_patches)
    HIP_TRY_FREE(hipMalloc((void**)&b->d_bg84, (size_t)SF_BG_COUNT * SF_OUT * SF_OUT));
    HIP_TRY_FREE(hipMemset(b->d_bg84, 0, (size_t)SF_BG_COUNT * SF_OUT * SF_OUT));
    HIP_TRY_FREE(hipMalloc((void**)&b->d_bg, (size_t)SF_BG_COUNT * SF_BG_STRIDE));
    HIP_TRY_FREE(hipMemset(b->d_bg, 0, (size_t)SF_BG_COUNT * SF_BG_STRIDE));
    {
      std::vector<double> arcs(86 * 8);
      std::vector<uint8_t> fa(36 * 256);
      int rc = sf_arc_table(arcs.data());
      for (int k = 0; k < 36 && rc == SF_OK; k++) rc = sf_image_fort_alpha(k, fa.data() + 256 * k);
      if (rc != SF_OK) {
        sf_destroy(b);
        return rc;
      }
      HIP_TRY_FREE(hipMalloc((void**)&b->d_arcs, arcs.size() * sizeof(double)));
      HIP_TRY_FREE(hipMemcpy(b->d_arcs, arcs.data(), arcs.size() * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY_FREE(hipMalloc((void**)&b->d_falpha, fa.size()));
      HIP_TRY_FREE(hipMemcpy(b->d_falpha, fa.data(), fa.size(), hipMemcpyHostToDevice));
    }
    HIP_TRY_FREE(hipMalloc((void**)&b->d_tabs, tabs.size() * sizeof(uint32_t)));
    HIP_TRY_FREE(hipMemcpy(b->d_tabs, tabs.data(), tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    // the score text: the built-in glyph atlas (sf_glyphs.h) and the four static backgrounds made with it
    HIP_TRY_FREE(hipMalloc((void**)&b->d_glyphs, sizeof(SfGlyphAtlas)));
    sf_glyphs_default(&b->h_glyphs);
    {
      const int rc = bake_default_glyphs(b);
      if (rc != SF_OK) {
        sf_destroy(b);
        return rc;
      }
    }
  }

  SfKernelArgs& a = b->args;
  a.state = b->d_state;
  a.lanes = lanes;
  a.n_envs = p->n_envs;
  a.consts = b->d_consts;
  a.spawn = b->d_spawn;
  a.spawn_mask = (unsigned)(spawn_len - 1);
  a.action_keys = 0;
  for (int i = 0; i < n_actions; i++) a.action_keys |= (unsigned long long)(keys[i] & 0xF) << (4 * i);
  a.n_actions = n_actions;
  a.start_vx = preset.start_vx;  // libm values of the host, SRC/configs.cpp:43-44
  a.start_vy = preset.start_vy;
  {
    // Everything else of baseConfig (SRC/configs.cpp:3-49) is identical in the four presets and is
    // compiled into the kernels (namespace sfc); the scoring triple is the SHAPED template
    // argument.  Refuse to run if a preset ever disagrees with what was compiled in.
    const double rm = (double)(preset.missile_radius + preset.fortress_radius);
    const double rs = (double)(preset.shell_radius + preset.ship_radius);
    const bool same =
        preset.width == (int)sfc::width_d && preset.height == (int)sfc::height_d &&
        preset.game_time == sfc::game_time && preset.sector_size == sfc::sector_size &&
        preset.lock_time == sfc::lock_time && preset.vuln_time == sfc::vuln_time &&
        preset.vuln_threshold == sfc::vuln_threshold && preset.explode_duration == sfc::explode_duration &&
        preset.turn_speed == sfc::turn_speed && preset.missile_speed == sfc::missile_speed &&
        preset.shell_speed == sfc::shell_speed && preset.ship_accel == sfc::ship_accel &&
        rm * rm == sfc::missile_hit_r2 && rs * rs == sfc::shell_hit_r2 &&
        (double)preset.small_hex == sfc::ndist_a &&
        ((double)preset.big_hex - (double)preset.small_hex) / 2.0 == sfc::ndist_b && preset.miss_penalty == 0 &&
        (preset.shaped ? (preset.missile_penalty == 0.05 && preset.ship_death_penalty == 1 && preset.destroy_fortress == 1)
                       : (preset.missile_penalty == 2.0 && preset.ship_death_penalty == 100 &&
                          preset.destroy_fortress == 100));
#define SF_E(nx, ny, px, py) nx, ny, px, py,
    const double compiled_hex[48] = {SF_BIG_HEX_EDGES(SF_E) SF_SMALL_HEX_EDGES(SF_E)};
#undef SF_E
    const bool same_hex = preset.big_hex == 200 && preset.small_hex == 40 &&
                          memcmp(compiled_hex, consts.data() + SF_LDS_BIGHEX, sizeof(compiled_hex)) == 0;
    if (!same || !same_hex) {
      sf_set_error("sf_create: preset `%s' differs from the constants compiled into the kernels", p->gametype);
      sf_destroy(b);
      return SF_ERR_ARG;
    }
  }
  b->obs_mode = p->obs_type;
  const bool image = p->obs_type == SF_OBS_IMAGE || p->obs_type == SF_OBS_IMAGE_RAW;
  a.obs_type = image ? SF_OBS_NONE : p->obs_type;  // frames come from sf_render, after the step kernel
  a.obs_f64 = (p->flags & SF_FLAG_OBS_F64) ? 1 : 0;
  a.real_shell_count = (p->flags & SF_FLAG_REAL_SHELL_COUNT) ? 1 : 0;
  a.auto_reset = (p->flags & SF_FLAG_NO_AUTO_RESET) ? 0 : 1;
  a.ref_reset_obs = (p->flags & SF_FLAG_REF_RESET_OBS) ? 1 : 0;
  a.obs_dim = p->obs_type == SF_OBS_MONITORS ? 10 : ((p->obs_type == SF_OBS_NONE || image) ? 0 : 15 + preset.n_keys);
  static_assert(sfc::pb_width == (double)(int)(450 * .2) && sfc::pb_height == (double)(int)(460 * .2), "ENV:57-58");
  static_assert(sfc::max_ticks == (double)(sfc::game_time / sfc::tick_ms), "ENV:165");
  a.acc = b->d_acc;
  a.dbg = nullptr;
  a.hint = nullptr;
  a.act_out = nullptr;
  if (write_action_records(b, p->seed, 0u, nullptr) != SF_OK) {  // sf_step_sampled's default stream: (seed, lane, tick 0)
    sf_destroy(b);
    return SF_ERR_HIP;
  }
  if (image && !getenv("SFMI_NO_RENDER_ORDER")) {  // the render kernel's launch order (sf_render.hip: pick_env)
    HIP_TRY_FREE(hipMalloc((void**)&a.hint, (size_t)(lanes / 64) * sizeof(unsigned long long)));
    HIP_TRY_FREE(hipMemset(a.hint, 0, (size_t)(lanes / 64) * sizeof(unsigned long long)));
  }
#ifdef SF_STAMPS  // diagnostic build (tools/stamps.py): per-wave clock stamps, never in the product
  HIP_TRY_FREE(hipMalloc((void**)&a.dbg, (size_t)(lanes / 64) * 16 * sizeof(unsigned long long)));
  HIP_TRY_FREE(hipMemset(a.dbg, 0, (size_t)(lanes / 64) * 16 * sizeof(unsigned long long)));
#endif

  a.draw = nullptr;
  a.draw_pics = 0;
  HIP_TRY_FREE(sf_launch_reset(a, 1, (unsigned)p->spawn_skip, (unsigned)p->spawn_stride, nullptr, nullptr));
  HIP_TRY_FREE(hipStreamSynchronize(nullptr));
  if (image) {  // the render caches and pictures, and the draw records the step launches of this batch keep current
    if (ensure_render_resources(b, nullptr) != SF_OK) {
      sf_destroy(b);
      return SF_ERR_HIP;
    }
    a.draw = b->d_draw;
    a.draw_pics = b->d_xcache ? 1 : 0;
  }
#undef HIP_TRY_FREE
  *out = b;
  return SF_OK;
}

extern "C" int sf_destroy(sf_batch* b) {
  if (!b) return SF_OK;
  DeviceGuard guard(b->device);
  if (b->d_state) (void)hipFree(b->d_state);
  if (b->d_consts) (void)hipFree(b->d_consts);
  if (b->d_spawn) (void)hipFree(b->d_spawn);
  if (b->d_acc) (void)hipFree(b->d_acc);
  if (b->d_actrec) (void)hipFree(b->d_actrec);
  if (b->d_scratch) (void)hipFree(b->d_scratch);
  if (b->d_ms_pos) (void)hipFree(b->d_ms_pos);
  if (b->d_ms_ang) (void)hipFree(b->d_ms_ang);
  if (b->d_bg) (void)hipFree(b->d_bg);
  if (b->d_tabs) (void)hipFree(b->d_tabs);
  if (b->d_xcache) (void)hipFree(b->d_xcache);
  if (b->d_bg84) (void)hipFree(b->d_bg84);
  if (b->d_arcs) (void)hipFree(b->d_arcs);
  if (b->d_falpha) (void)hipFree(b->d_falpha);
  if (b->d_draw) (void)hipFree(b->d_draw);
  if (b->d_gbg) (void)hipFree(b->d_gbg);
  if (b->d_gtabs) (void)hipFree(b->d_gtabs);
  if (b->d_glyphs) (void)hipFree(b->d_glyphs);
  if (b->args.dbg) (void)hipFree(b->args.dbg);
  if (b->args.hint) (void)hipFree(b->args.hint);
  delete b;
  return SF_OK;
}

extern "C" int sf_set_render_order_hint(sf_batch* b, const uint64_t* words_host, int n_words) {
  if (!b || !words_host || !b->args.hint || n_words != (b->n_envs + 63) / 64) {
    sf_set_error("sf_set_render_order_hint: needs an image batch and ceil(n_envs / 64) words");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(b->args.hint, words_host, (size_t)n_words * sizeof(uint64_t), hipMemcpyHostToDevice));
  return SF_OK;
}

#ifdef SF_STAMPS
extern "C" int sf_debug_read(sf_batch* b, unsigned long long* host) {
  DeviceGuard guard(b->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host, b->args.dbg, (size_t)(b->args.lanes / 64) * 16 * sizeof(unsigned long long),
                    hipMemcpyDeviceToHost));
  return SF_OK;
}
#endif

extern "C" int sf_n_envs(const sf_batch* b) { return b ? b->n_envs : SF_ERR_ARG; }
extern "C" int sf_obs_dim(const sf_batch* b) {
  if (!b) return SF_ERR_ARG;
  if (b->obs_mode == SF_OBS_IMAGE) return SF_OUT * SF_OUT;
  if (b->obs_mode == SF_OBS_IMAGE_RAW) return b->g_w ? b->g_w * b->g_h : SF_IMG_W * SF_IMG_H;
  return b->args.obs_dim;
}
extern "C" int sf_n_actions(const sf_batch* b) { return b ? b->act_count : SF_ERR_ARG; }
extern "C" int sf_tick_ms(const sf_batch* b) { return b ? sfc::tick_ms : SF_ERR_ARG; }
extern "C" int sf_max_ticks(const sf_batch* b) { return b ? (int)sfc::max_ticks : SF_ERR_ARG; }

static bool is_image(const sf_batch* b) { return b->obs_mode == SF_OBS_IMAGE || b->obs_mode == SF_OBS_IMAGE_RAW; }

extern "C" int sf_image_size(const sf_batch* b, int32_t* width, int32_t* height) {
  if (!b || !width || !height) return SF_ERR_ARG;
  *width = b->g_w ? b->g_w : SF_IMG_W;
  *height = b->g_w ? b->g_h : SF_IMG_H;
  return SF_OK;
}

extern "C" int sf_image_geometry_is_default(const sf_batch* b) { return b && b->g_w == 0 ? 1 : 0; }

extern "C" int sf_set_image_geometry(sf_batch* b, double scale, double vp_x, double vp_y, double vp_w, double vp_h, double line_width) {
  if (!b) {
    sf_set_error("sf_set_image_geometry: null batch");
    return SF_ERR_ARG;
  }
  if (!(scale > 0) || !(vp_w > 0) || !(vp_h > 0) || !(line_width > 0) || !(vp_w * scale < 1e6) || !(vp_h * scale < 1e6)) {
    sf_set_error("sf_set_image_geometry: scale, viewport size and line width must be positive");
    return SF_ERR_ARG;
  }
  const int w = (int)(vp_w * scale), h = (int)(vp_h * scale);  // ENV:57-58
  DeviceGuard guard(b->device);
  HIP_TRY(hipDeviceSynchronize());
  if (vp_x == SF_VP_X && vp_y == SF_VP_Y && vp_w == 450.0 && vp_h == 460.0 && w == SF_IMG_W && h == SF_IMG_H && line_width == SF_LINE_W) {
    b->g_w = b->g_h = 0;  // the default geometry: the fast frame kernel, the built-in glyph atlas
    sf_glyphs_default(&b->h_glyphs);
    if (memcmp(&b->h_glyphs, &b->baked_glyphs, sizeof(SfGlyphAtlas)) != 0) return bake_default_glyphs(b);
    HIP_TRY(hipMemcpy(b->d_glyphs, &b->h_glyphs, sizeof(SfGlyphAtlas), hipMemcpyHostToDevice));
    return SF_OK;
  }
  if (w < SF_OUT || h < SF_OUT || w >= 3 * SF_OUT || h >= 3 * SF_OUT || (long)w * h > 49152) {
    sf_set_error("sf_set_image_geometry: a %d x %d surface (scale %g, viewport %g x %g): width and height must lie in [%d, %d] "
                 "with width * height <= 49152 (the 84x84 INTER_AREA image of the trainer is a shrink below threefold)",
                 w, h, scale, vp_w, vp_h, SF_OUT, 3 * SF_OUT - 1);
    return SF_ERR_ARG;
  }
  // cairo cuts an arc into Bezier segments by the tolerance over its DEVICE radius (cairo-arc.c: _arc_segments_needed): the
  // explosion's radius-7 circle is one segment per half up to 5.4 device pixels, and that is the form the renderer draws
  if (!((double)w / vp_w <= 0.75) || !((double)h / vp_h <= 0.75)) {
    sf_set_error("sf_set_image_geometry: %d x %d pixels for a %g x %g viewport: more than 0.75 pixels per unit is a close-up "
                 "this renderer does not draw (the explosion's circle takes more Bezier segments there)", w, h, vp_w, vp_h);
    return SF_ERR_ARG;
  }
  std::vector<uint8_t> bg(((size_t)w * h + 15) & ~(size_t)15, 0);  // (whole 16-byte pieces: the kernel copies it that way)
  int rc = sf_image_background_geom(w, h, vp_x, vp_y, vp_w, vp_h, line_width, bg.data());
  if (rc != SF_OK) return rc;
  std::vector<uint32_t> tabs(16 * SF_OUT, 0u);
  const int ssize[2] = {w, h};
  for (int ax = 0; ax < 2; ax++) {
    int32_t first[SF_OUT], count[SF_OUT];
    float alpha[SF_OUT * 4];
    rc = sf_resize_area_tab(ssize[ax], SF_OUT, first, count, alpha);
    if (rc != SF_OK) return rc;
    for (int i = 0; i < SF_OUT; i++) {
      uint32_t* t = &tabs[8 * (ax * SF_OUT + i)];
      t[0] = (uint32_t)first[i];
      t[1] = (uint32_t)count[i];
      memcpy(t + 2, alpha + 4 * i, 4 * sizeof(float));
      if (first[i] < 0 || count[i] < 1 || count[i] > 4 || first[i] + count[i] > ssize[ax]) {
        sf_set_error("sf_set_image_geometry: INTER_AREA table out of range");
        return SF_ERR_ARG;
      }
    }
  }
  uint8_t* nbg = nullptr;
  uint32_t* ntabs = nullptr;
  HIP_TRY(hipMalloc((void**)&nbg, bg.size()));
  if (hipMalloc((void**)&ntabs, tabs.size() * sizeof(uint32_t)) != hipSuccess || hipMemcpy(nbg, bg.data(), bg.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ntabs, tabs.data(), tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(nbg);
    if (ntabs) (void)hipFree(ntabs);
    sf_set_error("sf_set_image_geometry: device allocation / copy failed");
    return SF_ERR_HIP;
  }
  if (b->d_gbg) (void)hipFree(b->d_gbg);
  if (b->d_gtabs) (void)hipFree(b->d_gtabs);
  b->d_gbg = nbg;
  b->d_gtabs = ntabs;
  b->g_w = w;
  b->g_h = h;
  b->g_sx = (double)w / vp_w;
  b->g_sy = (double)h / vp_h;
  b->g_vx = vp_x;
  b->g_vy = vp_y;
  b->g_lw = line_width;
  // no glyph atlas is known for this geometry: the seven-segment fallback until sf_set_score_glyphs gives one
  memset(&b->h_glyphs, 0, sizeof(SfGlyphAtlas));
  HIP_TRY(hipMemcpy(b->d_glyphs, &b->h_glyphs, sizeof(SfGlyphAtlas), hipMemcpyHostToDevice));
  return SF_OK;
}

extern "C" int sf_set_score_glyphs(sf_batch* b, const sf_score_glyphs* layout, const uint8_t* alpha) {
  if (!b) {
    sf_set_error("sf_set_score_glyphs: null batch");
    return SF_ERR_ARG;
  }
  SfGlyphAtlas G;
  const int box[4] = {SF_TXT_BOX_X0, SF_TXT_BOX_Y0, SF_TXT_BOX_X1, SF_TXT_BOX_Y1};
  const int rc = sf_glyphs_pack(layout, alpha, b->g_w ? nullptr : box, &G);
  if (rc != SF_OK) return rc;
  DeviceGuard guard(b->device);
  HIP_TRY(hipDeviceSynchronize());
  b->h_glyphs = G;
  if (!b->g_w) return bake_default_glyphs(b);
  HIP_TRY(hipMemcpy(b->d_glyphs, &b->h_glyphs, sizeof(SfGlyphAtlas), hipMemcpyHostToDevice));
  return SF_OK;
}

extern "C" int sf_get_score_glyphs(const sf_batch* b, int32_t* has_atlas, sf_score_glyphs* layout, uint8_t* alpha, size_t alpha_bytes) {
  if (!b || !has_atlas) {
    sf_set_error("sf_get_score_glyphs: null argument");
    return SF_ERR_ARG;
  }
  const SfGlyphAtlas& G = b->h_glyphs;
  *has_atlas = G.gw ? 1 : 0;
  if (!G.gw) return SF_OK;
  const size_t need = (size_t)SF_GLYPH_CHARS * G.gw * G.gh;
  if (!layout || !alpha || alpha_bytes < need) {
    sf_set_error("sf_get_score_glyphs: need layout and %zu bytes for alpha", need);
    return SF_ERR_ARG;
  }
  layout->gw = G.gw;
  layout->gh = G.gh;
  layout->advance = G.advance;
  layout->y0 = G.y0;
  for (int i = 0; i < SF_GLYPH_CHARS * 10; i++) layout->x0[i / 10][i % 10] = G.x0[i];
  memcpy(alpha, G.alpha, need);
  return SF_OK;
}

static int render(sf_batch* b, int mode, uint8_t* frames_dev, size_t env_stride, hipStream_t stream,
                  const uint8_t* stack_done = nullptr, int stack_slot = 0, int stack_n = 1,
                  const uint8_t* stack_prev = nullptr) {
  const size_t frame = mode == SF_OBS_IMAGE ? (size_t)SF_OUT * SF_OUT
                                            : (b->g_w ? (size_t)b->g_w * b->g_h : (size_t)SF_IMG_W * SF_IMG_H);
  if (b->g_w) {  // another geometry than the default: the general renderer (sf_render_generic.hip), from the state
    if (env_stride == 0) env_stride = frame;
    if (env_stride < frame) {
      sf_set_error("image frames: env_stride %zu below the frame size %zu", env_stride, frame);
      return SF_ERR_ARG;
    }
    if (stack_prev) {
      sf_set_error("sf_render_shift is built for the default image geometry; use sf_render_stack with this one");
      return SF_ERR_ARG;
    }
    // (the general renderer writes an 84x84 frame in 32-bit words; the surface itself goes out with byte heads and tails)
    if (mode == SF_OBS_IMAGE && ((((uintptr_t)frames_dev) | env_stride) & 3) != 0) {
      sf_set_error("image frames of a non-default geometry must be 4-byte aligned and env_stride a multiple of 4");
      return SF_ERR_ARG;
    }
    SF_FLUSH_VIEW(b, stream);
    if (stack_done)  // `current_obs *= masks` for the finished envs, then the new frame into its slot
      HIP_TRY(sf_launch_stack_clear(frames_dev - (size_t)stack_slot * frame, (size_t)stack_n * frame, stack_done, b->n_envs, stream));
    HIP_TRY(sf_launch_render_generic(b->d_state, b->n_envs, b->g_w, b->g_h, b->g_sx, b->g_sy, b->g_vx, b->g_vy, b->g_lw, b->d_consts + SF_LDS_TRIG, b->d_arcs, b->d_gbg,
                                     b->d_gtabs, frames_dev, env_stride, mode == SF_OBS_IMAGE ? 1 : 0, b->d_glyphs, stream));
    return SF_OK;
  }
  if (env_stride == 0) env_stride = frame;
  if (((uintptr_t)frames_dev & 15) != 0 || env_stride < frame || (env_stride & (mode == SF_OBS_IMAGE ? 15 : 7)) != 0) {
    sf_set_error("image frames must be 16-byte aligned, env_stride >= the frame size and a multiple of %d",
                 mode == SF_OBS_IMAGE ? 16 : 8);
    return SF_ERR_ARG;
  }
  SF_FLUSH_VIEW(b, stream);
  {
    // first frame of a batch that steps with a symbolic observation: the caches, pictures and draw records (feature-only
    // batches never pay for them; image batches made them in sf_create)
    const int rc = ensure_render_resources(b, stream);
    if (rc != SF_OK) return rc;
  }
  if (!b->draw_current) {  // the state changed otherwise than through a step launch of an image batch: records from the state
    SfKernelArgs da = b->args;
    da.draw = b->d_draw;
    da.draw_pics = b->d_xcache ? 1 : 0;
    HIP_TRY(sf_launch_drawrec(da, stream));
    b->draw_current = b->args.draw != nullptr;  // (only an image batch's step launches keep them current from here on)
  }
  const unsigned char* fpatch = b->d_xcache ? b->d_xcache + (size_t)b->n_envs * SF_XC_BYTES : nullptr;
  HIP_TRY(sf_launch_render(b->d_state, b->d_draw, b->n_envs, b->d_bg, b->d_bg84, b->d_tabs, frames_dev, env_stride, b->d_xcache, fpatch,
                           mode == SF_OBS_IMAGE ? 1 : 0, stack_done, stack_slot, stack_n, stack_prev, b->args.hint,
                           fpatch ? fpatch + 36 * SF_FP_BYTES + SF_XC_BYTES : nullptr, b->d_consts + SF_LDS_TRIG, b->d_arcs, b->d_falpha, b->d_glyphs, stream));
  return SF_OK;
}

extern "C" int sf_draw_records(sf_batch* b, void* host, size_t bytes, int from_state) {
  static_assert(SF_DRAW_RECORD_BYTES == SF_DR_BYTES, "sfmi.h vs sf_drawrec.h");
  if (!b || !host || bytes != (size_t)b->n_envs * SF_DR_BYTES) {
    sf_set_error("sf_draw_records: need a batch and n_envs * %d bytes", SF_DR_BYTES);
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  HIP_TRY(hipDeviceSynchronize());
  if (from_state) {
    SF_FLUSH_VIEW(b, nullptr);
    const int rc = ensure_render_resources(b, nullptr);
    if (rc != SF_OK) return rc;
    SfKernelArgs da = b->args;
    da.draw = b->d_draw;
    da.draw_pics = b->d_xcache ? 1 : 0;
    HIP_TRY(sf_launch_drawrec(da, nullptr));
    b->draw_current = b->args.draw != nullptr;
  } else if (!b->d_draw) {
    sf_set_error("sf_draw_records: this batch has no draw records yet (they come with its first frame)");
    return SF_ERR_ARG;
  }
  // (the caller gets them env by env, whatever the device layout: sf_drawrec.h SF_DR_LAYOUT)
  std::vector<unsigned char> raw((size_t)b->args.lanes * SF_DR_BYTES);
  HIP_TRY(hipMemcpy(raw.data(), b->d_draw, raw.size(), hipMemcpyDeviceToHost));
  for (long e = 0; e < b->n_envs; e++)  // (a record is contiguous: header, positions, headings)
    memcpy((unsigned char*)host + (size_t)e * SF_DR_BYTES, raw.data() + (size_t)(e >> 6) * SF_DR_TILE_BYTES + (size_t)(e & 63) * SF_DR_LANE_STRIDE,
           SF_DR_BYTES);
  return SF_OK;
}

extern "C" int sf_render(sf_batch* b, int mode, uint8_t* frames_dev, size_t env_stride, void* stream) {
  if (!b || !frames_dev) {
    sf_set_error("sf_render: null batch or output");
    return SF_ERR_ARG;
  }
  if (mode != SF_OBS_IMAGE && mode != SF_OBS_IMAGE_RAW) {
    sf_set_error("sf_render: mode must be SF_OBS_IMAGE or SF_OBS_IMAGE_RAW (got %d)", mode);
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  return render(b, mode, frames_dev, env_stride, (hipStream_t)stream);
}

extern "C" int sf_render_stack(sf_batch* b, uint8_t* stack_dev, int num_stack, int slot, const uint8_t* done_dev, void* stream) {
  if (!b || !stack_dev || num_stack < 1 || slot < 0 || slot >= num_stack) {
    sf_set_error("sf_render_stack: need a batch, the stack and 0 <= slot < num_stack");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  const size_t frame = (size_t)SF_OUT * SF_OUT;
  return render(b, SF_OBS_IMAGE, stack_dev + (size_t)slot * frame, (size_t)num_stack * frame, (hipStream_t)stream, done_dev, slot,
                num_stack);
}

extern "C" int sf_render_shift(sf_batch* b, const uint8_t* prev_stack_dev, uint8_t* stack_dev, int num_stack,
                               const uint8_t* done_dev, void* stream) {
  if (!b || !prev_stack_dev || !stack_dev || num_stack < 1 || prev_stack_dev == stack_dev ||
      ((uintptr_t)prev_stack_dev & 15) != 0) {
    sf_set_error("sf_render_shift: need a batch and two different 16-byte aligned stacks");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  const size_t frame = (size_t)SF_OUT * SF_OUT;
  return render(b, SF_OBS_IMAGE, stack_dev + (size_t)(num_stack - 1) * frame, (size_t)num_stack * frame, (hipStream_t)stream,
                done_dev, num_stack - 1, num_stack, prev_stack_dev);
}

extern "C" int sf_frame_stack_clear(uint8_t* stack_dev, size_t bytes_per_env, const uint8_t* done_dev, int n_envs, void* stream) {
  if (!stack_dev || !done_dev || n_envs <= 0 || (bytes_per_env & 15) != 0 || ((uintptr_t)stack_dev & 15) != 0) {
    sf_set_error("sf_frame_stack_clear: need 16-byte aligned stack, bytes_per_env a multiple of 16, done_dev, n_envs > 0");
    return SF_ERR_ARG;
  }
  HIP_TRY(sf_launch_stack_clear(stack_dev, bytes_per_env, done_dev, n_envs, (hipStream_t)stream));
  return SF_OK;
}

extern "C" int sf_reset(sf_batch* b, void* obs_dev, void* stream) {
  if (!b) {
    sf_set_error("sf_reset: null batch");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  const bool image = is_image(b);
  b->mslots_dirty = false;  // new games everywhere: no missiles, whatever a caller wrote into the slot view
  b->draw_current = false;
  HIP_TRY(hipMemsetAsync(b->d_acc + SF_ACC_OVERFLOW, 0, sizeof(unsigned long long), (hipStream_t)stream));  // new games: nothing is wrapped
  HIP_TRY(sf_launch_reset(b->args, 0, 0, 0, image ? nullptr : obs_dev, (hipStream_t)stream));
  if (image && obs_dev) return render(b, b->obs_mode, (uint8_t*)obs_dev, 0, (hipStream_t)stream);
  return SF_OK;
}

extern "C" int sf_step(sf_batch* b, const void* actions_dev, int act_type, void* obs_dev, int32_t* reward_dev,
                       uint8_t* done_dev, uint8_t* info_dev, void* stream) {
  if (!b || !actions_dev) {
    sf_set_error("sf_step: null batch or actions");
    return SF_ERR_ARG;
  }
  if (act_type != SF_ACT_U8 && act_type != SF_ACT_I32 && act_type != SF_ACT_I64) {
    sf_set_error("sf_step: act_type must be 1, 4 or 8 (got %d)", act_type);
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  const bool image = is_image(b);
  SF_FLUSH_VIEW(b, stream);
  HIP_TRY(sf_launch_step(b->args, b->autoturn, b->preset.shaped != 0, actions_dev, act_type, image ? nullptr : obs_dev,
                         reward_dev, done_dev, info_dev, 1, false, (hipStream_t)stream));
  b->draw_current = b->args.draw != nullptr;  // (an image batch's step launch leaves the draw records of the new state)
  if (image && obs_dev) return render(b, b->obs_mode, (uint8_t*)obs_dev, 0, (hipStream_t)stream);
  return SF_OK;
}

int sf_step_with_norm_partials(sf_batch* b, const void* actions_dev, int act_type, void* obs_dev, int32_t* reward_dev,
                               uint8_t* done_dev, uint8_t* info_dev, double* partials, double* ret, double gamma, int* rows_out,
                               void* stream) {
  if (!b || !actions_dev || !partials) {
    sf_set_error("sf_step_normalize: null batch, actions or normalizer");
    return SF_ERR_ARG;
  }
  if (act_type != SF_ACT_U8 && act_type != SF_ACT_I32 && act_type != SF_ACT_I64) {
    sf_set_error("sf_step_normalize: act_type must be 1, 4 or 8 (got %d)", act_type);
    return SF_ERR_ARG;
  }
  if (is_image(b) || b->args.obs_dim < 4) {
    sf_set_error("sf_step_normalize: VecNormalize applies to 1-D observations (rl/train.py:35)");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  SfKernelArgs args = b->args;
  args.n_partials = partials;
  args.n_ret = ret;
  args.n_gamma = gamma;
  SF_FLUSH_VIEW(b, stream);
  HIP_TRY(sf_launch_step(args, b->autoturn, b->preset.shaped != 0, actions_dev, act_type, obs_dev, reward_dev, done_dev,
                         info_dev, 1, false, (hipStream_t)stream));
  b->draw_current = b->args.draw != nullptr;  // (an image batch's step launch leaves the draw records of the new state)
  *rows_out = (int)(b->args.lanes / 64);
  return SF_OK;
}

extern "C" int sf_step_record(sf_batch* b, const void* actions_dev, int act_type, void* obs_dev, int32_t* reward_dev,
                              uint8_t* done_dev, uint8_t* info_dev, float* reward_f32, float* mask_f32, float* episode_rewards,
                              float* final_rewards, int64_t* actions_out, void* stream) {
  if (!b || !actions_dev || !reward_f32) {
    sf_set_error("sf_step_record: null batch, actions or reward_f32");
    return SF_ERR_ARG;
  }
  if (act_type != SF_ACT_U8 && act_type != SF_ACT_I32 && act_type != SF_ACT_I64) {
    sf_set_error("sf_step_record: act_type must be 1, 4 or 8 (got %d)", act_type);
    return SF_ERR_ARG;
  }
  if (((uintptr_t)reward_f32 | (uintptr_t)mask_f32 | (uintptr_t)episode_rewards | (uintptr_t)final_rewards) & 3 ||
      ((uintptr_t)actions_out & 7)) {
    sf_set_error("sf_step_record: float outputs must be 4-byte, actions_out 8-byte aligned");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  SfKernelArgs args = b->args;
  args.t_reward = reward_f32;
  args.t_mask = mask_f32;
  args.t_episode = episode_rewards;
  args.t_final = final_rewards;
  args.t_actions = (long long*)actions_out;
  const bool image = is_image(b);
  SF_FLUSH_VIEW(b, stream);
  HIP_TRY(sf_launch_step(args, b->autoturn, b->preset.shaped != 0, actions_dev, act_type, image ? nullptr : obs_dev,
                         reward_dev, done_dev, info_dev, 1, false, (hipStream_t)stream));
  b->draw_current = b->args.draw != nullptr;  // (an image batch's step launch leaves the draw records of the new state)
  if (image && obs_dev) return render(b, b->obs_mode, (uint8_t*)obs_dev, 0, (hipStream_t)stream);
  return SF_OK;
}

extern "C" int sf_rollout(sf_batch* b, const void* actions_dev, int act_type, int n_steps, void* obs_dev,
                          int32_t* reward_dev, uint8_t* done_dev, uint8_t* info_dev, void* stream) {
  if (!b || !actions_dev) {
    sf_set_error("sf_rollout: null batch or actions");
    return SF_ERR_ARG;
  }
  if (act_type != SF_ACT_U8 && act_type != SF_ACT_I32 && act_type != SF_ACT_I64) {
    sf_set_error("sf_rollout: act_type must be 1, 4 or 8 (got %d)", act_type);
    return SF_ERR_ARG;
  }
  if (n_steps <= 0 || (double)n_steps * b->n_envs * 8.0 >= 4294967296.0) {
    sf_set_error("sf_rollout: n_steps must be positive and n_steps * n_envs * 8 < 2^32 (got %d)", n_steps);
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  SF_FLUSH_VIEW(b, stream);
  if (is_image(b) && obs_dev) {
    // frames are rendered from the state in HBM, which the fused launch keeps in registers: with frames asked for, the K ticks
    // go out as K step launches, each followed by its frames -- what K sf_step calls do, in one call
    const size_t n = (size_t)b->n_envs, frame = (size_t)sf_obs_dim(b);
    for (int t = 0; t < n_steps; t++) {
      const size_t row = (size_t)t * n;
      SfKernelArgs args = b->args;
      if (args.events) args.events += row;  // (a rollout's event masks are [n_steps][n_envs], like its other outputs)
      HIP_TRY(sf_launch_step(args, b->autoturn, b->preset.shaped != 0, (const unsigned char*)actions_dev + row * (size_t)act_type,
                             act_type, nullptr, reward_dev ? reward_dev + row : nullptr, done_dev ? done_dev + row : nullptr,
                             info_dev ? info_dev + row : nullptr, 1, false, (hipStream_t)stream));
      b->draw_current = b->args.draw != nullptr;
      const int rc = render(b, b->obs_mode, (uint8_t*)obs_dev + row * frame, 0, (hipStream_t)stream);
      if (rc != SF_OK) return rc;
    }
    return SF_OK;
  }
  HIP_TRY(sf_launch_step(b->args, b->autoturn, b->preset.shaped != 0, actions_dev, act_type, obs_dev, reward_dev, done_dev, info_dev,
                         n_steps, true, (hipStream_t)stream));
  b->draw_current = b->args.draw != nullptr;  // (an image batch's step launch leaves the draw records of the new state)
  return SF_OK;
}

extern "C" int sf_seed_actions(sf_batch* b, uint64_t seed, uint32_t first_lane, void* stream) {
  if (!b) {
    sf_set_error("sf_seed_actions: null batch");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  return write_action_records(b, seed, first_lane, (hipStream_t)stream);
}

extern "C" int sf_step_sampled(sf_batch* b, uint8_t* actions_out_dev, void* obs_dev, int32_t* reward_dev, uint8_t* done_dev,
                               uint8_t* info_dev, void* stream) {
  if (!b) {
    sf_set_error("sf_step_sampled: null batch");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  const bool image = is_image(b);
  SfKernelArgs args = b->args;
  args.act_out = actions_out_dev;
  SF_FLUSH_VIEW(b, stream);
  HIP_TRY(sf_launch_step(args, b->autoturn, b->preset.shaped != 0, b->d_actrec, SF_ACT_SAMPLED, image ? nullptr : obs_dev,
                         reward_dev, done_dev, info_dev, 1, false, (hipStream_t)stream));
  b->draw_current = b->args.draw != nullptr;  // (an image batch's step launch leaves the draw records of the new state)
  if (image && obs_dev) return render(b, b->obs_mode, (uint8_t*)obs_dev, 0, (hipStream_t)stream);
  return SF_OK;
}

extern "C" int sf_rollout_sampled(sf_batch* b, int n_steps, uint8_t* actions_out_dev, void* obs_dev, int32_t* reward_dev,
                                  uint8_t* done_dev, uint8_t* info_dev, void* stream) {
  if (!b) {
    sf_set_error("sf_rollout_sampled: null batch");
    return SF_ERR_ARG;
  }
  if (n_steps <= 0 || (double)n_steps * b->n_envs >= 4294967296.0) {  // (no action array here: the outputs' 32-bit row offsets bound it)
    sf_set_error("sf_rollout_sampled: n_steps must be positive and n_steps * n_envs < 2^32 (got %d)", n_steps);
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  SfKernelArgs args = b->args;
  args.act_out = actions_out_dev;
  SF_FLUSH_VIEW(b, stream);
  if (is_image(b) && obs_dev) {
    // with frames: n_steps sampled step launches, each followed by its frames (sf_rollout does the same); the tiles' tick
    // counters move on by one per launch, so the actions drawn are the fused launch's
    const size_t n = (size_t)b->n_envs, frame = (size_t)sf_obs_dim(b);
    for (int t = 0; t < n_steps; t++) {
      const size_t row = (size_t)t * n;
      SfKernelArgs at = b->args;
      at.act_out = actions_out_dev ? actions_out_dev + row : nullptr;
      if (at.events) at.events += row;
      HIP_TRY(sf_launch_step(at, b->autoturn, b->preset.shaped != 0, b->d_actrec, SF_ACT_SAMPLED, nullptr,
                             reward_dev ? reward_dev + row : nullptr, done_dev ? done_dev + row : nullptr,
                             info_dev ? info_dev + row : nullptr, 1, false, (hipStream_t)stream));
      b->draw_current = b->args.draw != nullptr;
      const int rc = render(b, b->obs_mode, (uint8_t*)obs_dev + row * frame, 0, (hipStream_t)stream);
      if (rc != SF_OK) return rc;
    }
    return SF_OK;
  }
  HIP_TRY(sf_launch_step(args, b->autoturn, b->preset.shaped != 0, b->d_actrec, SF_ACT_SAMPLED, obs_dev, reward_dev, done_dev,
                         info_dev, n_steps, true, (hipStream_t)stream));
  b->draw_current = b->args.draw != nullptr;  // (an image batch's step launch leaves the draw records of the new state)
  return SF_OK;
}

extern "C" int sf_check_state(sf_batch* b, void* stream) {
  if (!b) return SF_ERR_ARG;
  DeviceGuard guard(b->device);
  unsigned long long two[2] = {0, 0};
  static_assert(SF_ACC_HANDOVER == SF_ACC_OVERFLOW + 1, "one copy for both");
  HIP_TRY(hipMemcpyAsync(two, b->d_acc + SF_ACC_OVERFLOW, sizeof(two), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  const unsigned long long bad = two[0];
  if (two[1]) {  // (sticky for the batch's life: its state cannot be trusted)
    sf_set_error("%llu times a wave of a split step launch gave up waiting for its tile's other wave: the state of this batch is "
                 "not the reference's any more (an internal error: please report it with the batch size and the GPU)", two[1]);
    return SF_ERR_STATE;
  }
  if (bad) {  // sticky: the fields stay wrapped until new games start (sf_reset clears the count)
    sf_set_error("%llu times since the last sf_reset a per-episode counter or timer left its packed width (a batch without "
                 "auto-reset stepped for several episodes without sf_reset): stats / timers of those envs have wrapped", bad);
    return SF_ERR_STATE;
  }
  return SF_OK;
}

extern "C" int sf_set_event_output(sf_batch* b, uint32_t* events_dev) {
  if (!b) return SF_ERR_ARG;
  if (((uintptr_t)events_dev & 3) != 0) {
    sf_set_error("sf_set_event_output: the buffer must be 4-byte aligned");
    return SF_ERR_ARG;
  }
  b->args.events = events_dev;
  return SF_OK;
}

extern "C" int sf_check_actions(sf_batch* b, void* stream) {
  if (!b) return SF_ERR_ARG;
  DeviceGuard guard(b->device);
  unsigned long long bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, b->d_acc + SF_EPISODE_STATS_LEN, sizeof(bad), hipMemcpyDeviceToHost,
                         (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if (bad) {
    HIP_TRY(hipMemsetAsync(b->d_acc + SF_EPISODE_STATS_LEN, 0, sizeof(bad), (hipStream_t)stream));
    sf_set_error("%llu action indices were outside [0, %d) and ran as NOOP", bad, b->act_count);
    return SF_ERR_ACTION;
  }
  return SF_OK;
}

extern "C" int sf_episode_stats(sf_batch* b, int64_t* out, int clear, void* stream) {
  if (!b || !out) return SF_ERR_ARG;
  DeviceGuard guard(b->device);
  // (all the words in one copy: the statistics, and behind them the sticky error counters -- a split launch whose hand-over
  //  timed out has played wrong games, and whoever reads statistics learns of it here without asking sf_check_state)
  unsigned long long all[SF_ACC_WORDS];
  HIP_TRY(hipMemcpyAsync(all, b->d_acc, sizeof(all), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  memcpy(out, all, SF_EPISODE_STATS_LEN * sizeof(int64_t));  // (delivered either way)
  if (all[SF_ACC_HANDOVER]) {  // ... but NOT cleared: the window's statistics stay where they are for whoever looks into this
    sf_set_error("%llu times a wave of a split step launch gave up waiting for its tile's other wave: the state of this batch is "
                 "not the reference's any more (an internal error: please report it with the batch size and the GPU).  `out` holds "
                 "the statistics; nothing was cleared; the counter is sticky: make a new batch", all[SF_ACC_HANDOVER]);
    return SF_ERR_STATE;
  }
  if (clear)
    HIP_TRY(hipMemcpyAsync(b->d_acc, kAccInit, SF_EPISODE_STATS_LEN * sizeof(int64_t), hipMemcpyHostToDevice,
                           (hipStream_t)stream));
  return SF_OK;
}

// ---- field access -------------------------------------------------------------------------

extern "C" int sf_n_fields(void) { return SF_F_COUNT; }

extern "C" int sf_field_info(int f, sf_field_desc* out) {
  if (f < 0 || f >= SF_F_COUNT || !out) {
    sf_set_error("sf_field_info: bad field id %d", f);
    return SF_ERR_FIELD;
  }
  out->name = sfl::kFields[f].name;
  out->elem_size = sfl::kFields[f].elem_size;
  out->count = sfl::kFields[f].count;
  out->is_float = sfl::kFields[f].is_float;
  return SF_OK;
}

extern "C" int sf_field_id(const char* name) {
  if (!name) return SF_ERR_FIELD;
  for (int f = 0; f < SF_F_COUNT; f++)
    if (!strcmp(name, sfl::kFields[f].name)) return f;
  sf_set_error("unknown field `%s'", name);
  return SF_ERR_FIELD;
}

// The tiled device layout never leaves the library: a small kernel gathers the field into (or
// scatters it from) a linear [count][n_envs] staging buffer, which is what the host sees.
static int field_copy(sf_batch* b, int f, void* host, size_t bytes, bool to_host) {
  if (!b || !host) return SF_ERR_ARG;
  if (f < 0 || f >= SF_F_COUNT) {
    sf_set_error("bad field id %d", f);
    return SF_ERR_FIELD;
  }
  const sfl::FieldMeta& m = sfl::kFields[f];
  const size_t total = (size_t)b->n_envs * m.elem_size * m.count;
  if (bytes != total) {
    sf_set_error("field %s: expected %zu bytes, got %zu", m.name, total, bytes);
    return SF_ERR_FIELD;
  }
  if (!to_host && (m.kind == SF_FK_BITS || m.kind == SF_FK_STATS)) {
    // these fields live in bit fields of packed words (sf_layout.h: SF_W_*): a value that does not fit is an error, not a
    // silent truncation (the reference keeps plain ints, SRC/game.hh:29-43)
    const int32_t* v = (const int32_t*)host;
    const long n = b->n_envs;
    auto fits = [](long x, int bits, int sgn) { return sgn ? (x >= -(1l << (bits - 1)) && x < (1l << (bits - 1))) : (x >= 0 && x < (1l << bits)); };
    if (m.kind == SF_FK_BITS) {
      const sfl::BitField bf = sfl::bit_field(f);
      const bool uns32 = m.elem_size == 4 && !bf.is_signed;
      for (long e = 0; e < n; e++) {
        const long x = uns32 ? (long)(uint32_t)v[e] : (long)v[e];
        if (!fits(x, bf.bits, bf.is_signed)) {
          sf_set_error("sf_set_field(%s): value %ld of env %ld does not fit the field's %d bits", m.name, x, e, bf.bits);
          return SF_ERR_ARG;
        }
      }
    } else {
      static const int kStatBits[SF_NSTAT] = {8, 8, 8, 10, 16, 8, 16, 16, 16, 16, 16, 12, 12};  // sf_layout.h: SF_W_*
      for (int k = 0; k < SF_NSTAT; k++)
        for (long e = 0; e < n; e++)
          if (!fits(v[(long)k * n + e], kStatBits[k], 0)) {
            sf_set_error("sf_set_field(stats): stats[%d] = %d of env %ld does not fit its %d bits", k, v[(long)k * n + e], e,
                         kStatBits[k]);
            return SF_ERR_ARG;
          }
      for (long e = 0; e < n; e++)
        if (v[3 * n + e] != v[e] + v[n + e] + v[2 * n + e]) {  // killShip's three call sites (SRC/game.cpp:339,345,413)
          sf_set_error("sf_set_field(stats): ship deaths (stats[3] = %d) of env %ld must be the sum of the big-hex, small-hex "
                       "and shell deaths (%d): it is not stored separately", v[3 * n + e], e, v[e] + v[n + e] + v[2 * n + e]);
          return SF_ERR_ARG;
        }
    }
  }
  DeviceGuard guard(b->device);
  HIP_TRY(hipDeviceSynchronize());
  const bool mview = m.kind == SF_FK_MPOOL;                      // missile_x / missile_y / missile_angle
  const bool mmask_write = f == SF_F_missile_mask && !to_host;   // which slots hold a missile
  if (mview || mmask_write) {
    // the per-slot view of the missiles: made on first use, refreshed from the pools unless it already holds edits
    const size_t cells = (size_t)b->n_envs * SF_NSLOT;
    if (!b->d_ms_pos) {
      HIP_TRY(hipMalloc((void**)&b->d_ms_pos, cells * 16));
      HIP_TRY(hipMalloc((void**)&b->d_ms_ang, cells * sizeof(int32_t)));
    }
    if (!b->mslots_dirty) HIP_TRY(sf_launch_mpool_to_slots(b->d_state, b->n_envs, b->d_ms_pos, b->d_ms_ang, nullptr));
  }
  if (!to_host) {
    HIP_TRY(hipMemcpy(b->d_scratch, host, total, hipMemcpyHostToDevice));
    b->draw_current = false;
  }
  if (mview) {
    const int which = f == SF_F_missile_x ? 0 : (f == SF_F_missile_y ? 1 : 2);
    HIP_TRY(sf_launch_mslot_component(b->d_ms_pos, b->d_ms_ang, (long)b->n_envs * SF_NSLOT, which, b->d_scratch,
                                      to_host ? 1 : 0, nullptr));
    if (!to_host) b->mslots_dirty = true;
  } else {
    HIP_TRY(sf_launch_field_copy(b->d_state, b->n_envs, f, b->d_scratch, to_host ? 1 : 0, nullptr));
    if (mmask_write) b->mslots_dirty = true;  // the pools follow the masks
  }
  HIP_TRY(hipDeviceSynchronize());
  if (to_host) HIP_TRY(hipMemcpy(host, b->d_scratch, total, hipMemcpyDeviceToHost));
  // (a packed per-episode field rewritten with values that fit repairs THAT field; the sticky count of sf_check_state is not
  //  touched here -- other fields, other envs may have wrapped: sf_clear_state_errors, for a caller that has rewritten them all)
  return SF_OK;
}

extern "C" int sf_clear_state_errors(sf_batch* b) {
  if (!b) {
    sf_set_error("sf_clear_state_errors: null batch");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(b->d_acc + SF_ACC_OVERFLOW, 0, sizeof(unsigned long long)));
  return SF_OK;
}

// Diagnostics: move a known number of bytes with the step kernel's own access pattern (whole
// 16-byte chunks, 64-lane rows) so rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated.
extern "C" int sf_calibration_copy(sf_batch* b, int which, size_t* bytes_moved) {
  if (!b) return SF_ERR_ARG;
  const int groups[2] = {SF_G_shell_pos, SF_G_missile_pos};
  if (which < 0 || which > 1) {
    sf_set_error("sf_calibration_copy: which must be 0 or 1");
    return SF_ERR_ARG;
  }
  DeviceGuard guard(b->device);
  const int g = groups[which];
  static_assert(sfl::kGroups[SF_G_shell_pos].chunk == 16 && sfl::kGroups[SF_G_missile_pos].chunk == 16, "chunk");
  HIP_TRY(sf_launch_group_copy(b->d_state, b->n_envs, g, b->d_scratch, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  if (bytes_moved) *bytes_moved = (size_t)b->n_envs * sfl::kGroups[g].slots * 16;
  return SF_OK;
}

extern "C" int sf_get_field(sf_batch* b, int f, void* host, size_t bytes) {
  return field_copy(b, f, host, bytes, true);
}
extern "C" int sf_set_field(sf_batch* b, int f, const void* host, size_t bytes) {
  return field_copy(b, f, const_cast<void*>(host), bytes, false);
}

// One state field of every env into DEVICE memory, [count][n_envs] in the field's element type like sf_get_field's host
// layout, ordered on `stream` behind the steps issued there and without a synchronise or a PCIe copy: what on-device
// bookkeeping reads between two steps (spacefortress_amd/durations.py).  Not for the missile fields (their per-slot view is a
// host-side convenience made on demand: sf_get_field).
extern "C" int sf_get_field_dev(sf_batch* b, int f, void* dev, size_t bytes, void* stream) {
  if (!b || !dev || f < 0 || f >= SF_F_COUNT) {
    sf_set_error("sf_get_field_dev: bad argument (field %d)", f);
    return b && dev ? SF_ERR_FIELD : SF_ERR_ARG;
  }
  const sfl::FieldMeta& m = sfl::kFields[f];
  const size_t total = (size_t)b->n_envs * m.elem_size * m.count;
  if (bytes != total) {
    sf_set_error("field %s: expected %zu bytes, got %zu", m.name, total, bytes);
    return SF_ERR_FIELD;
  }
  if (m.kind == SF_FK_MPOOL) {
    sf_set_error("sf_get_field_dev: %s is a per-slot view of the tiles' missile pools: read it with sf_get_field", m.name);
    return SF_ERR_FIELD;
  }
  DeviceGuard guard(b->device);
  SF_FLUSH_VIEW(b, (hipStream_t)stream);
  HIP_TRY(sf_launch_field_copy(b->d_state, b->n_envs, f, (unsigned char*)dev, 1, (hipStream_t)stream));
  return SF_OK;
}
