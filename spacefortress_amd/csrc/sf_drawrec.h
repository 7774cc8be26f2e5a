// sf_drawrec.h -- the DRAW RECORD of an env: what the frame kernel (sf_render.hip) needs of it, render-ready.
//
// sf_render_kernel is one WAVE per env and its bound is instruction issue: whatever a frame decides from its env's state
// alone -- which background variant to start from, whether the cached pictures apply, where the ship's box lies relative to
// the fortress's, the score's and the bar's -- is the same value in all 64 lanes and used to cost a wave instruction per
// env (307 vector + 256 scalar instructions per frame before the first pixel, round 3).  The step kernel has that state in
// registers with one LANE per env: there the same decisions cost a 64th.  So the image instantiations of sf_step_kernel
// (and sf_drawrec_kernel, for a state that was changed any other way: reset, sf_set_field, a features batch that renders)
// leave, per env, SF_DR_BYTES of HBM:
//   * a 32-byte header of finished decisions, which the frame kernel reads with ONE scalar load (s_load_dwordx8: the
//     values arrive in SGPRs, uniform by construction -- no v_readfirstlane, no 64-bit lane masks for uniform booleans);
//   * SF_DR_OBJS positions of 16 bytes, (x, y) as float64 -- cairo transforms path points in double precision and rounds them
//     to 1/256 pixel: a float32 position would move a corner now and then --: the ship, (the fortress: unused), the 20 missile
//     slots; lane s of the frame kernel loads the position of the object its line belongs to.  A missile's entry is written
//     by whichever lane of the step kernel holds it in the tile's pool, at [owner][slot], as the pool is compacted;
//   * the missiles' headings, 20 x int16 (whole degrees), behind the positions (the ship's and the fortress's: header word 7).
// Shells keep their (position, velocity) in the state and are read from there by the 14 % of the frames that have one.
//
// What is drawn where follows SRC/draw.cpp:227-270 (see sf_render.hip); nothing here changes a pixel: the functions below
// are the frame kernel's own round-3 arithmetic, moved to where both kernels compile them from one source.
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "sf_layout.h"
#include "sf_raster.h"

#define SF_DR_HDR_BYTES 32
#define SF_DR_OBJ_BYTES 16
#define SF_DR_OBJ_SHIP 0
#define SF_DR_OBJ_FORT 1
#define SF_DR_OBJ_MISSILE0 2
#define SF_DR_OBJS (2 + SF_NSLOT)
#define SF_DR_ANGLES_OFF (SF_DR_HDR_BYTES + SF_DR_OBJS * SF_DR_OBJ_BYTES) /* 384: int16 heading of missile slot s at + 2 s */
#define SF_DR_BYTES (SF_DR_ANGLES_OFF + 48) /* 432 */
// In HBM the records of a wave tile (64 envs, sf_layout.h) form one block of SF_DR_TILE_BYTES; env lane l's 16-byte piece p
// (pieces 0, 1 = header words 0..3, 4..7; piece 2 + k = object k's transform) sits at l * SF_DR_LANE_STRIDE + p *
// SF_DR_PIECE_STRIDE.  SF_DR_LAYOUT 0: env by env (384 contiguous bytes per env).  The frame kernel is a wave per env: its
// two scalar loads and its lanes' transform loads then touch three or four 128-byte lines, 0.5 KB per frame.
// SF_DR_LAYOUT 1: rows of 64 x 16 bytes like the state's chunks -- every store of the step kernel (a lane per env) a
// coalesced 1 KiB row, but a frame then pulls up to 23 different lines, 2.9 KB: measured 45.5 against 42.9 us per 16 384
// frames, for 0 us on the step kernel (profiles/r04_render_versions.md).
#ifndef SF_DR_LAYOUT
#define SF_DR_LAYOUT 0
#endif
#define SF_DR_PIECES (2 + SF_DR_OBJS)
#define SF_DR_TILE_BYTES (64 * SF_DR_BYTES)
#if SF_DR_LAYOUT != 0
#error "the rows-of-64 layout of round 4's A/B is gone: the headings' array has no place in it"
#endif
#if SF_DR_LAYOUT == 0
#define SF_DR_LANE_STRIDE SF_DR_BYTES
#define SF_DR_PIECE_STRIDE 16
#else
#define SF_DR_LANE_STRIDE 16
#define SF_DR_PIECE_STRIDE 1024
#endif
#define SF_DR_PIECE_OBJ0 2
// the cache bits of the step kernel's record stores (aux of the buffer-store builtins): 0 = plain -- the header's two pieces
// and the ship's and the fortress's transforms are 64 contiguous bytes per env and merge in L2; 16 = sc1, write-through
// like the state's chunks: each piece then leaves as a partial line of its own (step launch 8.45 against 7.74 us)
#ifndef SF_DR_AUX
#define SF_DR_AUX 0
#endif

// header words
#define SF_DRW_SHIP_X 0  /* (float)ship_x, bits */
#define SF_DRW_SHIP_Y 1
#define SF_DRW_POINTS 2  /* (int)mScore.mPoints: drawScore takes an int (SRC/draw.cpp:190,266) */
#define SF_DRW_OBJMASK 3 /* bit 0: the live ship's strokes, bit 1: the fortress's strokes IN PLACE, bits 2..21: missile slots */
#define SF_DRW_SHELLS 4  /* bits 0..19: live shell slots; bits 24..31: first lane of the shells' strokes when they ride in the
                            top lanes of the missiles' range (SF_DRF_MERGE_SHELLS) */
#define SF_DRW_FLAGS 5
#define SF_DRW_SERIAL 6  /* the env's time (ms) when the record was made: diagnostics only */
#define SF_DRW_ANGLES 7  /* the ship's heading (low 16 bits, whole degrees) | the fortress's (high 16) */

// SF_DRW_FLAGS: bits 0..7 = index of the background the frame starts from (4 (1 + sector) + variant with the fortress's
// picture in it, else the variant: sf_raster.h SF_BG_COUNT); bits 8..11 = the bar's state (0..10 tenths, 11 kill-ready)
#define SF_DRF_BG(f) ((f) & 0xFFu)
#define SF_DRF_BAR(f) (((f) >> 8) & 0xFu)
#define SF_DRF_SHIP_ALIVE (1u << 12)
#define SF_DRF_FORT_ALIVE (1u << 13)
#define SF_DRF_NEAR_TEXT (1u << 14)   /* something drawn before the score touches its box: not baked, not the picture */
#define SF_DRF_NEAR_BAR (1u << 15)
#define SF_DRF_EX_TEXT (1u << 16)     /* ... the dead ship's explosion does (its box wider by the reach) */
#define SF_DRF_EX_BAR (1u << 17)
#define SF_DRF_OTHER_TEXT (1u << 18)  /* the live ship or a projectile within reach of the score's box */
#define SF_DRF_OTHER_BAR (1u << 19)
#define SF_DRF_FORT_EX_PATCH (1u << 20)  /* the destroyed fortress's explosion: restored from its picture */
#define SF_DRF_FORT_EX_PLACE (1u << 21)  /* ... drawn in place, between the ship and the missiles */
#define SF_DRF_BAKED_TEXT (1u << 22)  /* the background holds the score 0000000 (= bit 0 of the variant) */
#define SF_DRF_BAKED_BAR (1u << 23)   /* ... the empty bar (= bit 1) */
#define SF_DRF_MERGE_SHELLS (1u << 24)
#define SF_DRF_MISSILE19 (1u << 25)   /* the twentieth missile slot is live (its strokes have no lanes of their own) */

namespace sfd {

#ifdef __HIPCC__
#define SFD_FN __device__ __forceinline__
#else
#define SFD_FN inline
#endif

struct Box {  // pixel rectangle [x0, x1) x [y0, y1) of the 90x92 surface
  int x0, y0, x1, y1;
  SFD_FN void clear() { x0 = y0 = 1 << 20; x1 = y1 = -1; }
  SFD_FN bool empty() const { return x1 <= x0 || y1 <= y0; }
  SFD_FN void add(int ax0, int ay0, int ax1, int ay1) {
    x0 = x0 < ax0 ? x0 : ax0; y0 = y0 < ay0 ? y0 : ay0; x1 = x1 > ax1 ? x1 : ax1; y1 = y1 > ay1 ? y1 : ay1;
  }
  SFD_FN bool meets(const Box& o) const { return x0 < o.x1 && o.x0 < x1 && y0 < o.y1 && o.y0 < y1; }
};

// A destination pixel of the 84x84 image reads two adjacent source columns and up to three adjacent rows: a picture saved
// with its 84x84 part is good where nothing else is drawn within that reach of its box (sf_render.hip: out_box)
constexpr int kReachX = 1, kReachY = 2;
// a box that meets the score's or the bar's box, wider by the reach or not, has y0 < kHudTopRows or y1 > kHudBottomRows
constexpr int kHudTopRows = SF_TXT_BOX_Y1 + kReachY, kHudBottomRows = SF_BAR_BOX_Y0 - kReachY;
// the fortress's picture: 355 +- 37.5, 315 +- 37.5 user units, in pixels (sf_render.hip: fort_patch_copy)
constexpr int kFpX0 = 37, kFpX1 = 53, kFpY0 = 39, kFpY1 = 55;
static_assert(kFpX0 <= (355 - 37.5 - SF_VP_X) * SF_SCALE && kFpX1 >= (355 + 37.5 - SF_VP_X) * SF_SCALE &&
              kFpY0 <= (315 - 37.5 - SF_VP_Y) * SF_SCALE && kFpY1 >= (315 + 37.5 - SF_VP_Y) * SF_SCALE, "fortress box");

SFD_FN Box text_box() { return Box{SF_TXT_BOX_X0, SF_TXT_BOX_Y0, SF_TXT_BOX_X1, SF_TXT_BOX_Y1}; }
SFD_FN Box bar_box() { return Box{SF_BAR_BOX_X0, SF_BAR_BOX_Y0, SF_BAR_BOX_X1, SF_BAR_BOX_Y1}; }
SFD_FN Box widened(const Box& b) { return Box{b.x0 - kReachX, b.y0 - kReachY, b.x1 + kReachX, b.y1 + kReachY}; }

// everything an explosion draws lies within 63 + 1.5 user units of its centre
SFD_FN Box explosion_box(float cx, float cy) {
  const float gx = sfr::dev_x(cx), gy = sfr::dev_y(cy), ext = 64.5f * (float)SF_SCALE;
  Box b;
  b.x0 = (int)floorf(gx - ext);
  b.y0 = (int)floorf(gy - ext);
  b.x1 = (int)ceilf(gx + ext);
  b.y1 = (int)ceilf(gy + ext);
  b.x0 = b.x0 > 0 ? b.x0 : 0;
  b.y0 = b.y0 > 0 ? b.y0 : 0;
  b.x1 = b.x1 < SF_IMG_W ? b.x1 : SF_IMG_W;
  b.y1 = b.y1 < SF_IMG_H ? b.y1 : SF_IMG_H;
  return b;
}
// a box of half-extent `ext` pixels around a position (user units): the live ship (25.5 + 1.5 user units), and the
// conservative boxes of the projectiles' strokes (a missile: 25 + 1.5, a shell: 16 + 1.5; + 0.01 px)
SFD_FN Box around(float x, float y, float ext) {
  const float gx = sfr::dev_x(x), gy = sfr::dev_y(y);
  return Box{(int)floorf(gx - ext), (int)floorf(gy - ext), (int)ceilf(gx + ext), (int)ceilf(gy + ext)};
}
constexpr float kShipExt = 27.f * (float)SF_SCALE;
constexpr float kMissileExt = 26.5f * (float)SF_SCALE + 0.01f;
constexpr float kShellExt = 17.5f * (float)SF_SCALE + 0.01f;

// drawVlner's state (SRC/draw.cpp:205-225,268): 0..10 tenths in grey .66, 11 = full and white (kill-ready)
SFD_FN int bar_state(int vlner, int fort_vuln_timer) {
  const bool kill = vlner > 10 && fort_vuln_timer < sfc::vuln_time;
  return kill ? 11 : (vlner > 10 ? 10 : vlner);
}

// Where a projectile's strokes can come near the score or the bar: bit 0 on the score's box, 1 on the bar's, 2 / 3 within
// reach of them.  A stroke lies within `ext` pixels of its object's position, so this is never false where the exact test
// on the stroke's own box (round 3) was true; where it is true without need the frame takes the general path to the same
// pixels (the score / the bar drawn in place instead of copied).
SFD_FN unsigned hud_flags_near(float x, float y, float ext) {
  const float gy = sfr::dev_y(y);
  if (!(gy - ext < (float)kHudTopRows || gy + ext > (float)kHudBottomRows)) return 0u;  // (all but never)
  const Box b = around(x, y, ext), t = text_box(), r = bar_box();
  return (b.meets(t) ? 1u : 0u) | (b.meets(r) ? 2u : 0u) | (b.meets(widened(t)) ? 4u : 0u) | (b.meets(widened(r)) ? 8u : 0u);
}
// ... the rows test alone, per lane and cheap: is hud_flags_near worth evaluating for this position
SFD_FN bool hud_rows_near(float y, float ext) {
  const float gy = sfr::dev_y(y);
  return gy - ext < (float)kHudTopRows || gy + ext > (float)kHudBottomRows;
}

struct Header {
  unsigned w[8];
};

// A box by its float edges, before floor / ceil: for a pixel box o = [x0, x1) x [y0, y1) with integer corners,
//   Box{floor(a.x0), floor(a.y0), ceil(a.x1), ceil(a.y1)}.meets(o)  ==  a.x0 < o.x1 && o.x0 < a.x1 && a.y0 < o.y1 && o.y0 < a.y1
// (floor(a) < n <=> a < n and n < ceil(b) <=> n < b for an integer n), and clamping the left side to the surface changes
// nothing for an o that lies on it (max(., 0) < o.x1 with o.x1 > 0, o.x0 < min(., W) with o.x0 < W).  Eight float
// compares against constants where the boxes cost eight floor / ceil, conversions and clamps per test: the step kernel runs
// this once per env and tick, with one wave per SIMD and nothing to hide an instruction behind.
struct FBox {
  float x0, y0, x1, y1;
  SFD_FN bool meets(const Box& o) const { return x0 < (float)o.x1 && (float)o.x0 < x1 && y0 < (float)o.y1 && (float)o.y0 < y1; }
};

// The decisions of a frame (sf_render.hip, round 3: between "round trip 2" and the barrier), from the env's state.
//   proj  = OR of hud_flags_near over the env's live missiles and shells
//   pics  = the batch has its pictures (the 36 fortress headings baked into backgrounds, the destroyed fortress's
//           explosion, the score / bar pictures): not with SFMI_NO_EXPLOSION_CACHE, which draws everything in place
SFD_FN Header make_header(double sx, double sy, int ship_angle, bool ship_alive, bool fort_alive, int fort_angle, float points, int vlner,
                          int fort_vuln_timer, unsigned mmask, unsigned smask, unsigned proj, bool pics, int time_ms) {
  Header h;
  const float ship_x = (float)sx, ship_y = (float)sy;
  // what was drawn before the fortress: the live ship (within 25.5 + 1.5 user units of its position) or the dead one's
  // explosion (63 + 1.5) -- sfd::around / sfd::explosion_box, as float edges
  const float gx = sfr::dev_x(ship_x), gy = sfr::dev_y(ship_y), ext = ship_alive ? kShipExt : 64.5f * (float)SF_SCALE;
  const FBox sb{gx - ext, gy - ext, gx + ext, gy + ext};
  const int sector = fort_angle / 10;
  // the fortress's picture is good when nothing the ship drew comes within reach of its box (+ what its 84x84 pixels read)
  const bool fort_pic = pics && fort_alive && fort_angle >= 0 && fort_angle < 360 && sector * 10 == fort_angle &&
                        !sb.meets(widened(Box{kFpX0, kFpY0, kFpX1, kFpY1}));
  // within reach of the score's / the bar's box.  The dead ship's explosion: its 84x84 pixels, recomputed or restored from
  // the cache, read that far and must not depend on whether the score / bar were baked in (round 3 widened the explosion's
  // box instead of the score's: the same test).  The live ship stays inside the big hexagon, rows 12.2 .. 81.6 +- 5.4 px:
  // never on the bar, but close.
  const bool mt = sb.meets(widened(text_box())), mb = sb.meets(widened(bar_box()));
  const bool ex_text = !ship_alive && mt, ex_bar = !ship_alive && mb;
  const bool other_text = (ship_alive && mt) || (proj & 4u), other_bar = (ship_alive && mb) || (proj & 8u);
  // (a live ship ON one of the boxes: not in play -- it dies outside the big hexagon -- but a caller's sf_set_field can put it there)
  const bool near_text = ex_text || (proj & 1u) || (ship_alive && sb.meets(text_box()));
  const bool near_bar = ex_bar || (proj & 2u) || (ship_alive && sb.meets(bar_box()));
  const int pnts = (int)points;
  const bool baked_text = pnts == 0 && !near_text, baked_bar = vlner == 0 && !near_bar;
  const unsigned variant = (baked_text ? 1u : 0u) | (baked_bar ? 2u : 0u);
  const unsigned bg = (fort_pic ? 4u * (1u + (unsigned)sector) : 0u) + variant;
  // the destroyed fortress explodes for 1000 ms where it stands: restored from its picture when what the ship drew stays
  // clear of it (wider by the reach), else drawn in place between the ship and the missiles
  const bool fe_clear = !sb.meets(widened(explosion_box((float)sfc::fort_x, (float)sfc::fort_y)));
  const bool fe_patch = !fort_alive && pics && fe_clear, fe_place = !fort_alive && !fe_patch;
  // shells: lane 4 s + k of the frame kernel draws stroke k of slot s -- unless the top lanes of the missiles' range are
  // free for them (1 + highest live slot <= 8, and the missile slots whose lanes those are empty: nearly always): then the
  // shells' strokes sit there, behind the missiles' as in the draw order, and go through draw_strokes with everything else
  constexpr int kFirstMissileLane = 7;  // lanes 0 .. 2: the ship's strokes, 3 .. 6: the fortress's, 7 ..: three per missile slot
  const int sh_hi = smask ? 32 - __builtin_clz(smask) : 0;  // 1 + highest live slot
  const int sh_base = 64 - 4 * sh_hi;
  const bool merge = smask != 0u && sh_hi <= 8 && (mmask >> ((sh_base - kFirstMissileLane) / 3)) == 0u;
  const bool fort_strokes = fort_alive && !fort_pic;
  h.w[SF_DRW_SHIP_X] = __builtin_bit_cast(unsigned, ship_x);
  h.w[SF_DRW_SHIP_Y] = __builtin_bit_cast(unsigned, ship_y);
  h.w[SF_DRW_POINTS] = (unsigned)pnts;
  h.w[SF_DRW_OBJMASK] = (ship_alive ? 1u : 0u) | (fort_strokes ? 2u : 0u) | ((mmask & SF_MASK_LOW) << SF_DR_OBJ_MISSILE0);
  h.w[SF_DRW_SHELLS] = (smask & SF_MASK_LOW) | (merge ? (unsigned)sh_base << 24 : 0u);
  h.w[SF_DRW_FLAGS] = bg | ((unsigned)bar_state(vlner, fort_vuln_timer) << 8) | (ship_alive ? SF_DRF_SHIP_ALIVE : 0u) |
                      (fort_alive ? SF_DRF_FORT_ALIVE : 0u) | (near_text ? SF_DRF_NEAR_TEXT : 0u) | (near_bar ? SF_DRF_NEAR_BAR : 0u) |
                      (ex_text ? SF_DRF_EX_TEXT : 0u) | (ex_bar ? SF_DRF_EX_BAR : 0u) | (other_text ? SF_DRF_OTHER_TEXT : 0u) |
                      (other_bar ? SF_DRF_OTHER_BAR : 0u) | (fe_patch ? SF_DRF_FORT_EX_PATCH : 0u) | (fe_place ? SF_DRF_FORT_EX_PLACE : 0u) |
                      (baked_text ? SF_DRF_BAKED_TEXT : 0u) | (baked_bar ? SF_DRF_BAKED_BAR : 0u) | (merge ? SF_DRF_MERGE_SHELLS : 0u) |
                      (((mmask >> 19) & 1u) ? SF_DRF_MISSILE19 : 0u);
  h.w[SF_DRW_SERIAL] = (unsigned)time_ms;
  h.w[SF_DRW_ANGLES] = ((unsigned)ship_angle & 0xFFFFu) | ((unsigned)fort_angle << 16);
  return h;
}

}  // namespace sfd
