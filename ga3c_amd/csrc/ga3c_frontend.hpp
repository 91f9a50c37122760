// Frame front-end on the GPU: RGB Atari frame -> the uint8 84x84 plane of ga3c/Environment.py:52-60, pushed into a
// device-resident 4-deep frame queue per agent (Environment.py:62-74), so that only the raw frame ever crosses
// PCIe and predictions read their [84,84,4] states out of HBM.
//
//   gray  = fma(b, .114, fma(g, .587, r * .299))  in f64      (np.dot's evaluation order, oracle/frame_frontend.py)
//   u8    = trunc(clip((gray - min) * (255 / (max - min)), 0, 255) + .5)   per-frame min / max (scipy bytescale)
//   plane = Pillow BILINEAR resize: horizontal pass then vertical pass, 22-bit fixed-point taps, 8-bit intermediate
//   queue : one uint32 per pixel, byte c = plane c, oldest first -> push = (word >> 8) | (plane << 24); the words
//           ARE the [84,84,4] uint8 HWC state the conv kernels consume.
//
// One workgroup (1024 threads) per frame.  Byte work bounded by memory, so every byte moves once: a thread loads
// its share of the frame -- up to FE_MAXG groups of 4 pixels = 12 (RGB) or 16 (RGBA) contiguous bytes each, all
// loads issued back to back -- straight into registers, keeps the f64 gray values there across the block-wide
// min/max reduction, and writes the 8-bit image into LDS (33.6 KB).  The horizontal pass gives each thread one
// fixed group of 4 output columns, so its taps sit in registers while it walks down the rows; the vertical pass
// reads 4 pixels per LDS dword.  The only other HBM traffic is the queue's read-modify-write (28 KB each way).
// Algorithmic bytes per frame: H*W*C in + 2 * 28,224 queue (+ 7,056 when the plane itself is asked for).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ga3c {

constexpr int FE_THREADS = 1024;
constexpr int FE_PRECISION_BITS = 32 - 8 - 2;

struct FrameArgs {
  const uint8_t* rgb;        // [n][H][W][C], device-visible; or, with src_off, the base the offsets count from
  const int64_t* src_off;    // [n] byte offset of each frame from rgb (frames scattered over transport slots); may be nullptr
  uint8_t* ring;             // [max_agents][hist][OH*OW] plane history for training rows; nullptr when hist == 0
  const int32_t* ring_slot;  // [n] history slot each plane goes to
  int hist;
  const int32_t* agents;     // [n] queue to push each plane into; nullptr = stateless (planes only)
  const uint8_t* reset;      // [n] non-zero = clear that agent's queue first (Environment.reset); may be nullptr
  uint8_t* planes;           // [n][OH*OW] or nullptr
  uint32_t* stacks;          // [max_agents][OH*OW]
  const int32_t *hb, *hk, *vb, *vk;   // resample tables (ga3c_resample.hpp): bounds [out][2], taps [out][ksize]
  int H, W, C, OH, OW, hks, vks;
};

__device__ __forceinline__ double fe_gray(uint32_t r, uint32_t g, uint32_t b) {
  return __fma_rn((double)b, 0.114, __fma_rn((double)g, 0.587, __dmul_rn((double)r, 0.299)));
}

__device__ __forceinline__ uint32_t fe_clip8(int32_t acc) {
  int32_t v = acc >> FE_PRECISION_BITS;
  // Keep the shift and the clamp apart: fused, ROCm 7.2's gfx950 backend pairs two of these into v_ashr_pk_u8_i32,
  // whose result it then ORs with further bytes as if bits 16..31 were zero -- they are not (seen as wrong pixels
  // 2 and 3 of every packed dword; tests/test_gpu_frontend.py catches it).
  asm volatile("" : "+v"(v));
  return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

constexpr int FE_MAXG = 10;        // pixel groups per thread: frames up to 4 * 10 * 1024 = 40,960 pixels (250 x 160 fits)
constexpr int FE_MAXK = 8;         // taps per output the register-resident horizontal pass holds (downscale <= 3.5x)
static_assert(FE_MAXK == 8, "the horizontal pass reads exactly two shifted dwords of source bytes per output");

// 4 consecutive pixels' r, g, b out of their 12 (C = 3) or 16 (C = 4) contiguous bytes
template <int C>
__device__ __forceinline__ void fe_load4(const uint8_t* p, double (&g)[4]) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
  if (C == 3) {
    const uint32_t a = w[0], b = w[1], c = w[2];
    g[0] = fe_gray(a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
    g[1] = fe_gray(a >> 24, b & 255u, (b >> 8) & 255u);
    g[2] = fe_gray((b >> 16) & 255u, b >> 24, c & 255u);
    g[3] = fe_gray((c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = fe_gray(w[q] & 255u, (w[q] >> 8) & 255u, (w[q] >> 16) & 255u);
  }
}

template <int C>
__global__ __launch_bounds__(FE_THREADS) void frame_frontend_kernel(FrameArgs a) {
  extern __shared__ __align__(16) uint8_t fe_lds[];
  const int tid = threadIdx.x, f = blockIdx.x;
  const int npx = a.H * a.W, ngrp = npx / 4;            // npx % 4 == 0 (checked on the host)
  const int nout = a.OH * a.OW;
  // LDS map: [resample tables][g8: npx][tmp: H*OW].  The tables are fetched here, under the frame loads' latency.
  int32_t* t_vb = reinterpret_cast<int32_t*>(fe_lds);
  int32_t* t_vk = t_vb + a.OH * 2;
  int32_t* t_hb = t_vk + a.OH * a.vks;
  int32_t* t_hk = t_hb + a.OW * 2;
  const int tab_bytes = ((a.OH * (2 + a.vks) + a.OW * (2 + a.hks)) * 4 + 15) & ~15;
  uint8_t* g8 = fe_lds + tab_bytes;
  uint8_t* tmp = g8 + ((npx + 15) & ~15);
  __shared__ double red_min[FE_THREADS / 64], red_max[FE_THREADS / 64];

  for (int i = tid; i < a.OH * 2; i += FE_THREADS) t_vb[i] = a.vb[i];
  for (int i = tid; i < a.OH * a.vks; i += FE_THREADS) t_vk[i] = a.vk[i];
  for (int i = tid; i < a.OW * 2; i += FE_THREADS) t_hb[i] = a.hb[i];
  for (int i = tid; i < a.OW * a.hks; i += FE_THREADS) t_hk[i] = a.hk[i];

  // ---- pass 1: the frame's bytes -> f64 gray in registers; per-frame min / max
  const uint8_t* src = a.src_off ? a.rgb + a.src_off[f] : a.rgb + (size_t)f * npx * C;
  double gray[FE_MAXG][4];
  double lo = 1e300, hi = -1e300;
#pragma unroll
  for (int j = 0; j < FE_MAXG; ++j) {
    const int grp = tid + j * FE_THREADS;
    if (grp < ngrp) {
      fe_load4<C>(src + (size_t)grp * 4 * C, gray[j]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        lo = fmin(lo, gray[j][q]);
        hi = fmax(hi, gray[j][q]);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, o));
    hi = fmax(hi, __shfl_xor(hi, o));
  }
  if ((tid & 63) == 0) {
    red_min[tid >> 6] = lo;
    red_max[tid >> 6] = hi;
  }
  __syncthreads();
  lo = red_min[0];
  hi = red_max[0];
#pragma unroll
  for (int w = 1; w < FE_THREADS / 64; ++w) {
    lo = fmin(lo, red_min[w]);
    hi = fmax(hi, red_max[w]);
  }
  double cscale = __dsub_rn(hi, lo);
  if (cscale == 0.0) cscale = 1.0;
  const double scale = 255.0 / cscale;

  // ---- pass 2: bytescale -> 8-bit gray image in LDS, 4 pixels per dword store
#pragma unroll
  for (int j = 0; j < FE_MAXG; ++j) {
    const int grp = tid + j * FE_THREADS;
    if (grp < ngrp) {
      uint32_t packed = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // numpy clips to [0, 255], adds .5 and truncates; inside that range adding first and clamping the integer is the
        // same value, outside it both give 0 or 255 -- one integer clamp instead of two f64 compare / select pairs
        const double t = __dmul_rn(__dsub_rn(gray[j][q], lo), scale);
        int v = (int)__dadd_rn(t, 0.5);
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        packed |= (uint32_t)v << (8 * q);
      }
      reinterpret_cast<uint32_t*>(g8)[grp] = packed;
    }
  }
  __syncthreads();

  // ---- pass 3: horizontal resample g8[H][W] -> tmp[H][OW].  Thread = one group of 4 output columns (taps and
  //      bounds in registers), walking down the rows `lanes_y` apart.
  const uint8_t* hsrc = g8;
  int cur_w = a.W;
  const int groups = a.OW / 4;
  if (a.W != a.OW) {
    const int lanes_y = FE_THREADS / groups;
    const int grp = tid % groups, y0 = tid / groups;
    int xmin[4], tap[4][FE_MAXK];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int xx = grp * 4 + q, n = t_hb[xx * 2 + 1];
      xmin[q] = t_hb[xx * 2];
#pragma unroll
      for (int k = 0; k < FE_MAXK; ++k) tap[q][k] = (k < a.hks && k < n) ? t_hk[xx * a.hks + k] : 0;
    }
    if (y0 < lanes_y) {
      for (int y = y0; y < a.H; y += lanes_y) {
        uint32_t packed = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // the <= 8 source bytes of this output as three ALIGNED LDS dwords, shifted into place (byte-granular
          // LDS reads would be merged by the compiler into wide reads at odd addresses); taps past the count are 0,
          // so whatever lies behind the last real tap is multiplied away
          const int idx = y * a.W + xmin[q];
          const uint32_t* w = reinterpret_cast<const uint32_t*>(g8 + (idx & ~3));
          const uint32_t sh = (uint32_t)(idx & 3) * 8;
          const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
          const uint32_t b03 = __funnelshift_r(w0, w1, sh), b47 = __funnelshift_r(w1, w2, sh);
          int32_t acc = 1 << (FE_PRECISION_BITS - 1);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc += (int32_t)((b03 >> (8 * k)) & 255u) * tap[q][k];
          acc += (int32_t)(b47 & 255u) * tap[q][4];
          if (a.hks > 5) {                                   // uniform: 160 -> 84 has 5 taps per output
#pragma unroll
            for (int k = 1; k < 4; ++k) acc += (int32_t)((b47 >> (8 * k)) & 255u) * tap[q][4 + k];
          }
          packed |= fe_clip8(acc) << (8 * q);
        }
        reinterpret_cast<uint32_t*>(tmp)[y * groups + grp] = packed;
      }
    }
    __syncthreads();
    hsrc = tmp;
    cur_w = a.OW;
  }

  // ---- pass 4: vertical resample -> plane[OH][OW]; push into the agent's queue
  const int agent = a.agents ? a.agents[f] : -1;
  const bool clear = a.reset && a.reset[f];
  uint32_t* stack = agent >= 0 ? a.stacks + (size_t)agent * nout : nullptr;
  uint8_t* plane = a.planes ? a.planes + (size_t)f * nout : nullptr;
  uint8_t* hist = (a.ring && agent >= 0) ? a.ring + ((size_t)agent * a.hist + a.ring_slot[f]) * nout : nullptr;
  for (int i = tid; i < a.OH * groups; i += FE_THREADS) {
    const int yy = i / groups, x0 = (i - yy * groups) * 4;
    const int p0 = yy * a.OW + x0;
    uint4 s = make_uint4(0, 0, 0, 0);
    if (stack && !clear) s = *reinterpret_cast<const uint4*>(stack + p0);   // issued before the taps are walked
    uint32_t v4[4];
    if (a.H != a.OH) {
      const int ymin = t_vb[yy * 2], n = t_vb[yy * 2 + 1];
      int32_t acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = 1 << (FE_PRECISION_BITS - 1);
      for (int k = 0; k < n; ++k) {
        const uint32_t w4 = *reinterpret_cast<const uint32_t*>(hsrc + (ymin + k) * cur_w + x0);
        const int32_t c = t_vk[yy * a.vks + k];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += (int32_t)((w4 >> (8 * q)) & 255u) * c;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v4[q] = fe_clip8(acc[q]);
    } else {
      const uint32_t w4 = *reinterpret_cast<const uint32_t*>(hsrc + yy * cur_w + x0);
#pragma unroll
      for (int q = 0; q < 4; ++q) v4[q] = (w4 >> (8 * q)) & 255u;
    }
    const uint32_t px4 = v4[0] | (v4[1] << 8) | (v4[2] << 16) | (v4[3] << 24);
    if (plane) *reinterpret_cast<uint32_t*>(plane + p0) = px4;
    if (hist) *reinterpret_cast<uint32_t*>(hist + p0) = px4;
    if (stack) {
      s.x = (s.x >> 8) | (v4[0] << 24);
      s.y = (s.y >> 8) | (v4[1] << 24);
      s.z = (s.z >> 8) | (v4[2] << 24);
      s.w = (s.w >> 8) | (v4[3] << 24);
      *reinterpret_cast<uint4*>(stack + p0) = s;
    }
  }
}

// The same push for frames that ARE planes already (C = 1, H x W = OH x OW: an actor that keeps the reference's gray /
// bytescale / resize on its side, or a source of ready-made planes): no arithmetic, only the frame queue kept on the device --
// the actor then ships 7,056 bytes per step instead of a 28,224-byte state, and rollout rows name states in the plane history.
__global__ __launch_bounds__(256) void plane_push_kernel(FrameArgs a) {
  const int f = blockIdx.x, nout = a.OH * a.OW;
  const uint8_t* src = a.src_off ? a.rgb + a.src_off[f] : a.rgb + (size_t)f * nout;
  const int agent = a.agents ? a.agents[f] : -1;
  const bool clear = a.reset && a.reset[f];
  uint32_t* stack = agent >= 0 ? a.stacks + (size_t)agent * nout : nullptr;
  uint8_t* plane = a.planes ? a.planes + (size_t)f * nout : nullptr;
  uint8_t* hist = (a.ring && agent >= 0) ? a.ring + ((size_t)agent * a.hist + a.ring_slot[f]) * nout : nullptr;
  // The plane lies in the transport's registered host segment: every load of it is a trip over PCIe.  All of a thread's loads
  // are requested before the first is used -- as one load per trip of the loop (round 3) the kernel paid seven bus round
  // trips in a row, 26.7 us per batch of 80 planes and the biggest kernel of the running engine (30 % of the GPU's time with
  // 256 agents, profiles/README.md, round 4) for 565 KB that the bus moves in 16 us.
  constexpr int TRIPS = 8;                                          // 8 x 1024 pixels >= 84 x 84
  for (int q0 = 0; q0 < nout; q0 += 4 * 256 * TRIPS) {
    uint32_t px[TRIPS];
    uint4 st[TRIPS];
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int p0 = q0 + 4 * (256 * k + (int)threadIdx.x);        // nout % 4 == 0 (checked on the host)
      px[k] = p0 < nout ? *reinterpret_cast<const uint32_t*>(src + p0) : 0u;
    }
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int p0 = q0 + 4 * (256 * k + (int)threadIdx.x);
      st[k] = (stack && !clear && p0 < nout) ? *reinterpret_cast<const uint4*>(stack + p0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < TRIPS; ++k) {
      const int p0 = q0 + 4 * (256 * k + (int)threadIdx.x);
      if (p0 >= nout) continue;
      const uint32_t px4 = px[k];
      if (plane) *reinterpret_cast<uint32_t*>(plane + p0) = px4;
      if (hist) *reinterpret_cast<uint32_t*>(hist + p0) = px4;
      if (stack) {
        uint4 s = st[k];
        s.x = (s.x >> 8) | ((px4 & 255u) << 24);
        s.y = (s.y >> 8) | (((px4 >> 8) & 255u) << 24);
        s.z = (s.z >> 8) | (((px4 >> 16) & 255u) << 24);
        s.w = (s.w >> 8) | ((px4 >> 24) << 24);
        *reinterpret_cast<uint4*>(stack + p0) = s;
      }
    }
  }
}

// Training rows out of the plane history: x[b] = the [OH,OW,4] uint8 state whose newest plane is history entry
// seq[b] of agent[b] (planes seq-3 .. seq, oldest first) -- what the agent's queue held right after that push.
// The rows' names travel with the launch (n > 0) instead of being read out of the pinned arrays -- every work item's loads
// waited for two scalar loads over PCIe in front of them -- and the batch's returns and actions ride along (SmallCopy: the last
// workgroup copies them out of pinned memory; as two hipMemcpyAsync they were two 5-us copy kernels on the train stream in
// front of every step: 7 % of the GPU's time in the running engine, profiles/README.md, round 4).
struct HistRows { int64_t seq[192]; int32_t agent[192]; int n; };

__global__ __launch_bounds__(256) void gather_history_kernel(const uint8_t* __restrict__ ring, const int32_t* __restrict__ agents,
                                                             const int64_t* __restrict__ seqs, int hist, int nout,
                                                             uint8_t* __restrict__ x, int B, SmallCopy sc, const HistRows hr) {
  if (blockIdx.x == gridDim.x - 1) small_copy(sc);
  const int groups = nout / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)B * groups; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / groups), g = (int)(i - (int64_t)b * groups);
    const int ag = hr.n ? hr.agent[b < 192 ? b : 0] : agents[b];
    const int64_t sq = hr.n ? hr.seq[b < 192 ? b : 0] : seqs[b];
    const uint8_t* base = ring + (size_t)ag * hist * nout;
    uint32_t w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = *reinterpret_cast<const uint32_t*>(base + (size_t)((sq - 3 + c) % hist) * nout + 4 * g);
    uint4 o;   // pixel q of the group: bytes (plane0, plane1, plane2, plane3)
    o.x = (w[0] & 255u) | ((w[1] & 255u) << 8) | ((w[2] & 255u) << 16) | ((w[3] & 255u) << 24);
    o.y = ((w[0] >> 8) & 255u) | (((w[1] >> 8) & 255u) << 8) | (((w[2] >> 8) & 255u) << 16) | (((w[3] >> 8) & 255u) << 24);
    o.z = ((w[0] >> 16) & 255u) | (((w[1] >> 16) & 255u) << 8) | (((w[2] >> 16) & 255u) << 16) | (((w[3] >> 16) & 255u) << 24);
    o.w = (w[0] >> 24) | ((w[1] >> 24) << 8) | ((w[2] >> 24) << 16) | ((w[3] >> 24) << 24);
    *reinterpret_cast<uint4*>(x + ((size_t)b * nout + 4 * g) * 4) = o;
  }
}

// bytes of dynamic LDS the kernel needs for a frame geometry
inline size_t frontend_lds_bytes(int H, int W, int OH, int OW, int hks, int vks) {
  const size_t tab = ((size_t)(OH * (2 + vks) + OW * (2 + hks)) * 4 + 15) & ~(size_t)15;
  const size_t g8 = ((size_t)H * W + 15) & ~(size_t)15;
  return tab + g8 + (((size_t)H * OW + 15) & ~(size_t)15);
}

}  // namespace ga3c
