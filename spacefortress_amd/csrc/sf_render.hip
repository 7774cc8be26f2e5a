// sf_render.hip -- the image observation: SSF_Env._draw (ENV:203-206) + WrapPyTorch.observation
// (rl/envs.py:28-30) for a whole batch, on the device.
//
// The reference renders every env with cairo into a 90x92 RGB24 surface (grayscale=True, line width
// 3 user units = 0.6 px, SRC/draw.cpp:257-270), takes the grey channel (cv2.cvtColor, an identity
// on R=G=B) and the trainer's wrapper shrinks the frame to 84x84 with cv2.INTER_AREA.
//
// Here: ONE WAVEFRONT PER ENV.  The frame lives in LDS as bytes (8.3 KB: 12+ waves per CU), starts
// as a copy of the static hexagon background, and every stroke of the reference's draw order is a
// convex quad (a line with butt caps, an arc chord, a filled rectangle) composited OVER it with
// 8-bit arithmetic, like the image backend does.  The quads of one draw phase are built one per
// lane (<= 64 per round); the wave then walks the live ones in order (ballot + readlane, no LDS
// list) and, for each, its lanes take the pixels of its bounding box: exact area coverage from an
// edge integral (no arrays, no scratch).  The epilogue resamples 90x92 -> 84x84 from LDS with the
// INTER_AREA tables (four output bytes per lane and store) or copies the raw frame out.
//
// Pixel values: what is drawn where, in which order and grey follows the reference; the
// anti-aliasing model is ours (cairo is not in this image).  Pixel parity with cairo + cv2 is
// UNPINNED; tests pin this kernel to the numpy restatement of the same model (oracle/render_np.py).
#include <hip/hip_runtime.h>

#include "sf_internal.h"
#include "sf_raster.h"

namespace {

constexpr int kFbBytes = SF_IMG_W * SF_IMG_H;          // 8280
constexpr int kFbWords = kFbBytes / 4;                 // 2070
constexpr int kFbPadWords = (SF_IMG_W * (SF_IMG_H + 1) + 3) / 4 + 1;  // one spare row for zero-weight taps
constexpr int kTabWords = 2 * SF_OUT * 4;              // x and y taps: {first, a0, a1, a2}

struct d2_t {
  double x, y;
};
struct i4_t {
  int x, y, z, w;
};

#define R_CHUNK(group, s) (tile + sfl::chunk_offset(SF_G_##group, (s)))
#define R_LD(T, base, off) (*reinterpret_cast<const T*>((base) + (off)))

struct Quad {
  float x[4], y[4];
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// mean over t in [0,1] of clamp(ya + t*(yb-ya), 0, 1)
__device__ __forceinline__ float ramp_mean(float ya, float yb) {
  float lo = fminf(ya, yb), d = fabsf(yb - ya);  // the mean does not depend on the direction
  if (d < 1e-6f) return clamp01(lo + 0.5f * d);
  const float inv = 1.0f / d;
  const float ta = clamp01(-lo * inv), tb = clamp01((1.0f - lo) * inv);
  return (1.0f - tb) + (tb - ta) * (lo + 0.5f * d * (ta + tb));
}

// area of quad /\ pixel [px,px+1]x[py,py+1]:  | sum over edges of the integral of clamp(y,0,1) dx |
__device__ __forceinline__ float quad_cover(const Quad& q, float px, float py) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int f = (e + 1) & 3;
    const float x0 = q.x[e] - px, y0 = q.y[e] - py, x1 = q.x[f] - px, y1 = q.y[f] - py;
    const float xa = clamp01(x0), xb = clamp01(x1);
    const float w = xb - xa;
    if (w != 0.f) {
      const float slope = (y1 - y0) / (x1 - x0);
      const float ya = y0 + (xa - x0) * slope, yb = y0 + (xb - x0) * slope;
      s += w * ramp_mean(ya, yb);
    }
  }
  return fabsf(s);
}

__device__ __forceinline__ float bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Composite this round's quads (one per lane, `valid` lanes only) in lane order.
__device__ __forceinline__ void draw_quads(uint8_t* fb, const Quad& mine, int grey, bool valid, int lane) {
  unsigned long long live = __ballot(valid);
  while (live) {
    const int src = __builtin_ctzll(live);
    live &= live - 1;
    Quad q;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      q.x[k] = bcast(mine.x[k], src);
      q.y[k] = bcast(mine.y[k], src);
    }
    const int c = __builtin_amdgcn_readlane(grey, src);
    const float fx0 = fminf(fminf(q.x[0], q.x[1]), fminf(q.x[2], q.x[3]));
    const float fx1 = fmaxf(fmaxf(q.x[0], q.x[1]), fmaxf(q.x[2], q.x[3]));
    const float fy0 = fminf(fminf(q.y[0], q.y[1]), fminf(q.y[2], q.y[3]));
    const float fy1 = fmaxf(fmaxf(q.y[0], q.y[1]), fmaxf(q.y[2], q.y[3]));
    // clip the bounding box to the surface (also rejects NaN / far-away geometry)
    if (!(fx1 > 0.f && fy1 > 0.f && fx0 < (float)SF_IMG_W && fy0 < (float)SF_IMG_H)) continue;
    const int bx0 = (int)floorf(fmaxf(fx0, 0.f)), by0 = (int)floorf(fmaxf(fy0, 0.f));
    const int bx1 = (int)ceilf(fminf(fx1, (float)SF_IMG_W)), by1 = (int)ceilf(fminf(fy1, (float)SF_IMG_H));
    const int bw = bx1 - bx0, n = bw * (by1 - by0);
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      if (i < n) {
        const int ry = i / bw, rx = i - ry * bw;
        const int px = bx0 + rx, py = by0 + ry;
        float area = quad_cover(q, (float)px, (float)py);
        area = fminf(area, 1.f);
        const int m = (int)(area * 255.f + 0.5f);
        if (m > 0) {
          uint8_t* p = fb + py * SF_IMG_W + px;
          *p = (uint8_t)sfr::over_un8(*p, c, m);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// user space -> device space (SRC/draw.cpp:259-260)
__device__ __forceinline__ float dev_x(float x) { return (x - (float)SF_VP_X) * (float)SF_SCALE; }
__device__ __forceinline__ float dev_y(float y) { return (y - (float)SF_VP_Y) * (float)SF_SCALE; }

// Stroke of the segment A-B (wireframe coordinates), butt caps, width SF_LINE_W, under
// translate(pos) rotate(angle) (drawWireFrame, SRC/draw.cpp:112-129)
__device__ __forceinline__ Quad line_quad(float ax, float ay, float bx, float by, float ca, float sa, float posx,
                                          float posy) {
  const float ux = bx - ax, uy = by - ay;
  const float inv = (float)(SF_LINE_W / 2) / sqrtf(ux * ux + uy * uy);
  const float nx = -uy * inv, ny = ux * inv;
  const float lx[4] = {ax + nx, bx + nx, bx - nx, ax - nx};
  const float ly[4] = {ay + ny, by + ny, by - ny, ay - ny};
  Quad q;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    q.x[k] = dev_x(posx + ca * lx[k] - sa * ly[k]);
    q.y[k] = dev_y(posy + sa * lx[k] + ca * ly[k]);
  }
  return q;
}

__device__ __forceinline__ Quad rect_quad(float x0, float y0, float x1, float y1) {
  Quad q;
  q.x[0] = dev_x(x0); q.y[0] = dev_y(y0);
  q.x[1] = dev_x(x1); q.y[1] = dev_y(y0);
  q.x[2] = dev_x(x1); q.y[2] = dev_y(y1);
  q.x[3] = dev_x(x0); q.y[3] = dev_y(y1);
  return q;
}

__device__ __forceinline__ void sincos_deg(float deg, float* s, float* c) {
  sincosf(deg * 0.017453292519943295f, s, c);
}

// wireframe segments (ax, ay, bx, by), SRC/wireframe.cpp:11-67
__constant__ float kShipLines[3][4] = {{-18, 0, 18, 0}, {-18, 18, 0, 0}, {0, 0, -18, -18}};
__constant__ float kFortLines[4][4] = {{0, 0, 36, 0}, {0, -18, 18, -18}, {18, -18, 18, 18}, {18, 18, 0, 18}};
__constant__ float kMissileLines[3][4] = {{0, 0, -25, 0}, {0, 0, -5, 5}, {0, 0, -5, -5}};
__constant__ float kShellLines[4][4] = {{-8, 0, 0, -6}, {0, -6, 16, 0}, {16, 0, 0, 6}, {0, 6, -8, 0}};

// seven-segment masks (bit 0 = A top, clockwise, bit 6 = G middle) for 0-9 and '-'
__constant__ unsigned char kSegs[11] = {0x3F, 0x06, 0x5B, 0x4F, 0x66, 0x6D, 0x7D, 0x07, 0x7F, 0x6F, 0x40};

// drawExplosion (SRC/draw.cpp:145-175): 7 rings of twelve 10-degree arcs, then a radius-7 circle.
// Piece p of 96: one chord quad per arc / per 30 degrees of the circle.
__device__ __forceinline__ Quad explosion_quad(int p, float cx, float cy, int* grey) {
  float r, a0, a1;
  if (p < 84) {
    const int ring = p / 12, k = p - ring * 12;
    const int radius = 15 + 8 * ring;
    a0 = (float)(30 * k + 3 * (ring + 1));
    a1 = a0 + 10.f;
    r = (float)radius;
    *grey = radius < 60 ? 191 : 128;  // .75 / .5
  } else {
    a0 = (float)(30 * (p - 84));
    a1 = a0 + 30.f;
    r = 7.f;
    *grey = 191;
  }
  float s0, c0, s1, c1;
  sincos_deg(a0, &s0, &c0);
  sincos_deg(a1, &s1, &c1);
  const float ri = r - (float)(SF_LINE_W / 2), ro = r + (float)(SF_LINE_W / 2);
  Quad q;
  q.x[0] = dev_x(cx + ri * c0); q.y[0] = dev_y(cy + ri * s0);
  q.x[1] = dev_x(cx + ro * c0); q.y[1] = dev_y(cy + ro * s0);
  q.x[2] = dev_x(cx + ro * c1); q.y[2] = dev_y(cy + ro * s1);
  q.x[3] = dev_x(cx + ri * c1); q.y[3] = dev_y(cy + ri * s1);
  return q;
}

__device__ __forceinline__ void draw_explosion(uint8_t* fb, float cx, float cy, int lane) {
  for (int round = 0; round < 2; round++) {
    const int p = round * 64 + lane;
    int grey = 0;
    const Quad q = explosion_quad(p < 96 ? p : 0, cx, cy, &grey);
    draw_quads(fb, q, grey, p < 96, lane);
  }
}

}  // namespace

struct SfRenderArgs {
  const unsigned char* state;
  int n_envs;
  const uint32_t* bg;    // kFbWords
  const uint32_t* tabs;  // kTabWords: x taps [84] then y taps [84], each {first, a0, a1, a2}
  uint8_t* out;
  int resize;            // 1: [n][84][84], 0: [n][92][90]
};

__global__ __launch_bounds__(64) void sf_render_kernel(SfRenderArgs a) {
  __shared__ uint32_t fbw[kFbPadWords];
  __shared__ uint32_t tabw[kTabWords];
  uint8_t* fb = reinterpret_cast<uint8_t*>(fbw);
  const int env = blockIdx.x, lane = threadIdx.x;

  for (int i = lane; i < kFbWords; i += 64) fbw[i] = a.bg[i];
  for (int i = kFbWords + lane; i < kFbPadWords; i += 64) fbw[i] = 0;
  if (a.resize)
    for (int i = lane; i < kTabWords; i += 64) tabw[i] = a.tabs[i];

  // this env's lane of its wave tile
  const unsigned char* tile = a.state + (long)(env >> 6) * sfl::kTileBytes;
  const int l = env & 63;
  const int o16 = l * 16, o8 = l * 8, o2 = l * 2;
  const d2_t sp = R_LD(d2_t, R_CHUNK(ship_pos, 0), o16);
  const i4_t tb = R_LD(i4_t, R_CHUNK(timers_b, 0), o16);
  const i4_t sc = R_LD(i4_t, R_CHUNK(score, 0), o16);
  const i4_t mi = R_LD(i4_t, R_CHUNK(misc, 0), o16);
  const int ship_angle = R_LD(int16_t, R_CHUNK(small, 0), o8);
  const int fort_angle = R_LD(int16_t, R_CHUNK(small, 0), o8 + 2);
  const unsigned flags = R_LD(uint8_t, R_CHUNK(small, 0), o8 + 6);
  const unsigned mmask = (unsigned)mi.z, smask = (unsigned)mi.w;
  const float points = __int_as_float(sc.x);
  const int vlner = sc.z;
  const int fort_vuln_timer = tb.w;
  __syncthreads();

  const float ship_x = (float)sp.x, ship_y = (float)sp.y;

  // ---- ship (SRC/draw.cpp:233-237)
  if (flags & SF_FL_SHIP_ALIVE) {
    float s, c;
    sincos_deg((float)ship_angle, &s, &c);
    const int k = lane < 3 ? lane : 0;
    const Quad q = line_quad(kShipLines[k][0], kShipLines[k][1], kShipLines[k][2], kShipLines[k][3], c, s, ship_x, ship_y);
    draw_quads(fb, q, 255, lane < 3, lane);
  } else {
    draw_explosion(fb, ship_x, ship_y, lane);
  }
  // ---- fortress (:238-242)
  if (flags & SF_FL_FORT_ALIVE) {
    float s, c;
    sincos_deg((float)fort_angle, &s, &c);
    const int k = lane < 4 ? lane : 0;
    const Quad q = line_quad(kFortLines[k][0], kFortLines[k][1], kFortLines[k][2], kFortLines[k][3], c, s,
                             (float)sfc::fort_x, (float)sfc::fort_y);
    draw_quads(fb, q, 255, lane < 4, lane);
  } else {
    draw_explosion(fb, (float)sfc::fort_x, (float)sfc::fort_y, lane);
  }
  // ---- missiles (:243-247): slot order, three segments each
  if (mmask) {
    const int slot = lane / 3, k = lane - slot * 3;
    const bool valid = lane < 3 * SF_NSLOT && ((mmask >> slot) & 1u);
    Quad q = {};
    if (valid) {
      const d2_t m = R_LD(d2_t, R_CHUNK(missile_pos, slot), o16);
      const int ang = R_LD(int16_t, R_CHUNK(missile_ang, slot), o2);
      float s, c;
      sincos_deg((float)ang, &s, &c);
      q = line_quad(kMissileLines[k][0], kMissileLines[k][1], kMissileLines[k][2], kMissileLines[k][3], c, s, (float)m.x,
                    (float)m.y);
    }
    draw_quads(fb, q, 255, valid, lane);
  }
  // ---- shells (:248-253): only once they are more than 21 away from the fortress
  if (smask) {
    for (int round = 0; round < 2; round++) {
      const int p = round * 64 + lane;
      const int slot = p >> 2, k = p & 3;
      bool valid = p < 4 * SF_NSLOT && ((smask >> slot) & 1u);
      Quad q = {};
      if (valid) {
        const d2_t s = R_LD(d2_t, R_CHUNK(shell_pos, slot), o16);
        const d2_t v = R_LD(d2_t, R_CHUNK(shell_vel, slot), o16);
        const double dx = s.x - sfc::fort_x, dy = s.y - sfc::fort_y;
        valid = sqrt(dx * dx + dy * dy) > 21.0;
        // mAngle = stdAngle(rad2deg(atan2(dy, dx))) at launch (SRC/game.cpp:263); the velocity kept in
        // the state has that direction.  drawWireFrame takes it as an int (truncation).
        double ang = atan2(v.y, v.x) * 180.0 / M_PI;
        if (ang < 0) ang += 360.0;
        float sn, cs;
        sincos_deg((float)(int)ang, &sn, &cs);
        q = line_quad(kShellLines[k][0], kShellLines[k][1], kShellLines[k][2], kShellLines[k][3], cs, sn, (float)s.x,
                      (float)s.y);
      }
      draw_quads(fb, q, 255, valid, lane);
    }
  }
  // ---- score (drawScore, :190-203): "%07d" of (int)points, grey .5, as seven-segment digits
  {
    int pnts = (int)points;
    const bool neg = pnts < 0;
    unsigned mag = neg ? (unsigned)(-(long)pnts) : (unsigned)pnts;
    // character d (0 = leftmost of 7): digits right-aligned, zero padded; a sign takes the first cell
    const int cell = lane / 7, seg = lane - cell * 7;
    unsigned div = 1;
    for (int i = 0; i < 6 - cell; i++) div *= 10;
    const int digit = (int)((mag / div) % 10);
    const int glyph = (neg && cell == 0) ? 10 : digit;
    const bool valid = lane < 49 && ((kSegs[glyph] >> seg) & 1);
    const float gx = SF_TXT_X0 + SF_TXT_ADV * cell + SF_TXT_PAD, gy = SF_TXT_TOP;
    const float W = SF_TXT_W, H = SF_TXT_H, T = SF_TXT_T, m0 = 0.5f * (SF_TXT_H - SF_TXT_T), m1 = 0.5f * (SF_TXT_H + SF_TXT_T);
    float x0, y0, x1, y1;
    switch (seg) {
      case 0: x0 = 0; x1 = W; y0 = 0; y1 = T; break;          // A top
      case 1: x0 = W - T; x1 = W; y0 = T; y1 = m0; break;     // B upper right
      case 2: x0 = W - T; x1 = W; y0 = m1; y1 = H - T; break; // C lower right
      case 3: x0 = 0; x1 = W; y0 = H - T; y1 = H; break;      // D bottom
      case 4: x0 = 0; x1 = T; y0 = m1; y1 = H - T; break;     // E lower left
      case 5: x0 = 0; x1 = T; y0 = T; y1 = m0; break;         // F upper left
      default: x0 = 0; x1 = W; y0 = m0; y1 = m1; break;       // G middle
    }
    const Quad q = rect_quad(gx + x0, gy + y0, gx + x1, gy + y1);
    draw_quads(fb, q, 128, valid, lane);
  }
  // ---- vulnerability bar (drawVlner, :205-225)
  {
    const bool kill = vlner > 10 && fort_vuln_timer < sfc::vuln_time;  // :268
    const int v = vlner > 10 ? 10 : vlner;
    const Quad q = lane == 0 ? rect_quad(255.f, 522.f, 455.f, 532.f) : rect_quad(255.f, 522.f, 255.f + 20.f * (float)v, 532.f);
    draw_quads(fb, q, lane == 0 ? 84 : (kill ? 255 : 168), lane == 0 || (lane == 1 && v > 0), lane);
  }
  __syncthreads();

  // ---- epilogue
  if (!a.resize) {
    uint32_t* out = reinterpret_cast<uint32_t*>(a.out + (size_t)env * kFbBytes);
    for (int i = lane; i < kFbWords; i += 64) out[i] = fbw[i];
    return;
  }
  const float* tabf = reinterpret_cast<const float*>(tabw);
  uint32_t* out = reinterpret_cast<uint32_t*>(a.out + (size_t)env * (SF_OUT * SF_OUT));
  for (int d = lane; d < SF_OUT * SF_OUT / 4; d += 64) {
    const int dy = d / (SF_OUT / 4), q4 = d - dy * (SF_OUT / 4);
    const int fy = (int)tabw[4 * (SF_OUT + dy)];
    const float b0 = tabf[4 * (SF_OUT + dy) + 1], b1 = tabf[4 * (SF_OUT + dy) + 2], b2 = tabf[4 * (SF_OUT + dy) + 3];
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int dx = 4 * q4 + j;
      const int fx = (int)tabw[4 * dx];
      const float a0 = tabf[4 * dx + 1], a1 = tabf[4 * dx + 2], a2 = tabf[4 * dx + 3];
      const uint8_t* r0 = fb + fy * SF_IMG_W + fx;
      const uint8_t* r1 = r0 + SF_IMG_W;
      const uint8_t* r2 = r1 + SF_IMG_W;
      // resizeArea_: per source row buf = sum alpha*S, then sum += beta*buf, in table order
      const float h0 = ((float)r0[0] * a0 + (float)r0[1] * a1) + (float)r0[2] * a2;
      const float h1 = ((float)r1[0] * a0 + (float)r1[1] * a1) + (float)r1[2] * a2;
      const float h2 = ((float)r2[0] * a0 + (float)r2[1] * a1) + (float)r2[2] * a2;
      const float sum = (b0 * h0 + b1 * h1) + b2 * h2;
      int v = (int)rintf(sum);  // saturate_cast<uchar>(float): round half to even, clamp
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      packed |= (uint32_t)v << (8 * j);
    }
    out[d] = packed;
  }
}

hipError_t sf_launch_render(const unsigned char* state, int n_envs, const uint32_t* bg, const uint32_t* tabs,
                            uint8_t* out, int resize, hipStream_t stream) {
  if (n_envs <= 0) return hipSuccess;
  SfRenderArgs a{state, n_envs, bg, tabs, out, resize};
  hipLaunchKernelGGL(sf_render_kernel, dim3((unsigned)n_envs), dim3(64), 0, stream, a);
  return hipGetLastError();
}
