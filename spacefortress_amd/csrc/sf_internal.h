// sf_internal.h -- declarations shared between the kernels, the host tables and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sf_glyphs.h"
#include "sf_layout.h"
#include "sfmi.h"

// sf_kernels.hip
hipError_t sf_launch_reset(const SfKernelArgs& a, int first, unsigned cursor0, unsigned stride, void* obs,
                           hipStream_t stream);
hipError_t sf_launch_step(const SfKernelArgs& a, bool autoturn, bool shaped, const void* actions, int act_type, void* obs,
                          int32_t* reward, uint8_t* done, uint8_t* info, int n_steps, bool fused, hipStream_t stream);

// the envs' draw records (sf_drawrec.h) from the state as it is: a.draw / a.draw_pics say where and for which pictures
hipError_t sf_launch_drawrec(const SfKernelArgs& a, hipStream_t stream);

// gather (to_linear) / scatter one field between the tiled state and a linear [count][n_envs] buffer
hipError_t sf_launch_field_copy(unsigned char* state, int n_envs, int field, unsigned char* linear, int to_linear,
                                hipStream_t stream);

// the missile fields as the reference has them, [SF_NSLOT][n_envs] slot-major ((x, y) pairs and int32 headings), from /
// to the tiles' pools; and one component of that view from / to a linear buffer (which: 0 x, 1 y, 2 heading as int16)
hipError_t sf_launch_mpool_to_slots(const unsigned char* state, int n_envs, void* sl_pos, int32_t* sl_ang, hipStream_t stream);
hipError_t sf_launch_slots_to_mpool(unsigned char* state, long lanes, int n_envs, const void* sl_pos, const int32_t* sl_ang,
                                    hipStream_t stream);
hipError_t sf_launch_mslot_component(void* sl_pos, int32_t* sl_ang, long total, int which, void* linear, int to_linear,
                                     hipStream_t stream);

hipError_t sf_launch_group_copy(const unsigned char* state, int n_envs, int group, unsigned char* linear,
                                hipStream_t stream);

// sf_render.hip: one wave per env; bg = 92*90 bytes, bg84 = its 84*84 INTER_AREA image, tabs = SF_TAB_WORDS
hipError_t sf_launch_render(const unsigned char* state, const unsigned char* draw, int n_envs, const uint32_t* bg, const uint32_t* bg84,
                            const uint32_t* tabs, uint8_t* out, size_t out_stride, unsigned char* xcache,
                            const unsigned char* fpatch, int resize, const uint8_t* stack_done, int stack_slot, int stack_n,
                            const uint8_t* stack_prev, const unsigned long long* hint, const unsigned char* hud,
                            const double* trig, const double* arcs, const unsigned char* falpha, const SfGlyphAtlas* glyphs,
                            hipStream_t stream);
// the score / bar pictures (SF_HUD_BYTES, sf_raster.h)
hipError_t sf_launch_hud_pictures(const uint32_t* bg, const uint32_t* bg84, const uint32_t* tabs, unsigned char* hud,
                                  const SfGlyphAtlas* glyphs, hipStream_t stream);
// ... and the 36 x 4 backgrounds with the fortress in them, behind the four plain ones (SF_BG_COUNT, sf_raster.h)
hipError_t sf_launch_fort_patches(uint32_t* bg, uint32_t* bg84, const uint32_t* tabs, unsigned char* fpatch, const double* arcs,
                                  const unsigned char* falpha, hipStream_t stream);

// sf_normalize.hip: reduce + apply (two launches); partials = SF_NORM_GROUPS x 2 (dim + 1) doubles; stats is the
// buffer of this step's parity, stats_next the other parity's
#define SF_NORM_GROUPS 256
hipError_t sf_launch_normalize(const void* obs, void* obs_out, int obs_f64, const int32_t* rew, float* rew_out, double* ret,
                               int n, int dim, double gamma, double eps, double clipob, double cliprew, int do_ob,
                               int do_ret, double* partials, const double* stats, double* stats_next,
                               hipStream_t stream);

// sf_render_generic.hip: the image observation in any geometry (sf_set_image_geometry): one workgroup per env, the W x H
// surface in dynamic LDS, bg = W * H bytes (the hexagons), tabs = the INTER_AREA taps (8 words per destination column, then
// per row: first, count, 4 weights, 2 pad)
hipError_t sf_launch_render_generic(const unsigned char* state, int n_envs, int W, int H, double sx, double sy, double vp_x, double vp_y,
                                    double line_w, const double* trig, const double* arcs, const uint8_t* bg, const uint32_t* tabs,
                                    uint8_t* out, size_t out_stride, int resize, const SfGlyphAtlas* glyphs, hipStream_t stream);
hipError_t sf_launch_stack_clear(uint8_t* stack, size_t bytes_per_env, const uint8_t* done, int n, hipStream_t stream);

hipError_t sf_launch_normalize_after_step(const void* obs, void* obs_out, int obs_f64, const int32_t* rew, float* rew_out, int n,
                                          int dim, double eps, double clipob, double cliprew, int do_ob, int do_ret,
                                          const double* partials, int rows, const double* stats, double* stats_next,
                                          hipStream_t stream);
// sf_capi.cpp: sf_step with the VecNormalize reduction riding on it (partials: [2 (obs_dim + 1)][lanes / 64])
int sf_step_with_norm_partials(sf_batch* b, const void* actions_dev, int act_type, void* obs_dev, int32_t* reward_dev,
                               uint8_t* done_dev, uint8_t* info_dev, double* partials, double* ret, double gamma, int* rows_out,
                               void* stream);

// sf_host.cpp (no HIP calls: usable and tested without a GPU)
void sf_host_fill_consts(const sf_preset& p, double* consts /* SF_CONST_DOUBLES */);
void sf_set_error(const char* fmt, ...);
// sf_image.cpp: the score text's glyph atlas (sf_glyphs.h) -- the built-in one, a caller's, and the static background with
// the score 0000000 of a given atlas baked in
void sf_glyphs_default(SfGlyphAtlas* G);
int sf_glyphs_pack(const sf_score_glyphs* layout, const uint8_t* alpha, const int* box, SfGlyphAtlas* G);
int sf_image_static_glyphs(int variant, const SfGlyphAtlas* G, uint8_t* out);
