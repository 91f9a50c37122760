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
// One workgroup (1024 threads) per frame.  Byte work bounded by memory: the 100,800 frame bytes are fetched ONCE
// (dword loads, coalesced) into LDS -- the min/max pass and the bytescale pass both read that copy -- the 8-bit
// gray image (33.6 KB) and the horizontal-pass image (17.6 KB, aliased onto the dead RGB copy) stay in LDS, and
// the only other HBM traffic is the queue's read-modify-write (28 KB each way).  Frames too big for that LDS
// budget (e.g. 250x160) re-read the RGB bytes from L2 instead (CACHE = false).
// Algorithmic bytes per frame: H*W*C in + 2 * 28,224 queue + 7,056 plane (when asked for).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ga3c {

constexpr int FE_THREADS = 1024;
constexpr int FE_PRECISION_BITS = 32 - 8 - 2;

struct FrameArgs {
  const uint8_t* rgb;        // [n][H][W][C], device-visible
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
  const int32_t v = acc >> FE_PRECISION_BITS;
  return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// gray of pixel i from a byte array (LDS copy or global memory)
template <typename P>
__device__ __forceinline__ double fe_gray_at(P bytes, int i, int C) {
  const int o = i * C;
  return fe_gray(bytes[o], bytes[o + 1], bytes[o + 2]);
}

template <bool CACHE>
__global__ __launch_bounds__(FE_THREADS) void frame_frontend_kernel(FrameArgs a) {
  extern __shared__ __align__(16) uint8_t fe_lds[];
  const int tid = threadIdx.x, f = blockIdx.x;
  const int npx = a.H * a.W, nbytes = npx * a.C;
  const int nout = a.OH * a.OW;
  // LDS map: [tables][g8: npx][scratch: max(CACHE ? nbytes : 0, H*OW)]
  int32_t* t_hb = reinterpret_cast<int32_t*>(fe_lds);
  int32_t* t_hk = t_hb + a.OW * 2;
  int32_t* t_vb = t_hk + a.OW * a.hks;
  int32_t* t_vk = t_vb + a.OH * 2;
  const int tab_bytes = ((a.OW * (2 + a.hks) + a.OH * (2 + a.vks)) * 4 + 15) & ~15;
  uint8_t* g8 = fe_lds + tab_bytes;
  uint8_t* scratch = g8 + ((npx + 15) & ~15);
  __shared__ double red_min[FE_THREADS / 64], red_max[FE_THREADS / 64];

  for (int i = tid; i < a.OW * 2; i += FE_THREADS) t_hb[i] = a.hb[i];
  for (int i = tid; i < a.OW * a.hks; i += FE_THREADS) t_hk[i] = a.hk[i];
  for (int i = tid; i < a.OH * 2; i += FE_THREADS) t_vb[i] = a.vb[i];
  for (int i = tid; i < a.OH * a.vks; i += FE_THREADS) t_vk[i] = a.vk[i];

  const uint8_t* src = a.rgb + (size_t)f * nbytes;
  if (CACHE) {   // nbytes % 4 == 0 and the frame base is 4-byte aligned (checked on the host)
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(scratch);
    for (int i = tid; i < nbytes / 4; i += FE_THREADS) d32[i] = s32[i];
    __syncthreads();
  }
  const uint8_t* px = CACHE ? scratch : src;

  // ---- pass 1: per-frame min / max of the f64 gray image
  double lo = 1e300, hi = -1e300;
  for (int i = tid; i < npx; i += FE_THREADS) {
    const double g = fe_gray_at(px, i, a.C);
    lo = g < lo ? g : lo;
    hi = g > hi ? g : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((tid & 63) == 0) {
    red_min[tid >> 6] = lo;
    red_max[tid >> 6] = hi;
  }
  __syncthreads();
  lo = red_min[0];
  hi = red_max[0];
  for (int w = 1; w < FE_THREADS / 64; ++w) {
    lo = red_min[w] < lo ? red_min[w] : lo;
    hi = red_max[w] > hi ? red_max[w] : hi;
  }
  double cscale = __dsub_rn(hi, lo);
  if (cscale == 0.0) cscale = 1.0;
  const double scale = 255.0 / cscale;

  // ---- pass 2: bytescale -> 8-bit gray image in LDS
  for (int i = tid; i < npx; i += FE_THREADS) {
    double t = __dmul_rn(__dsub_rn(fe_gray_at(px, i, a.C), lo), scale);
    t = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t);
    g8[i] = (uint8_t)(int)__dadd_rn(t, 0.5);
  }
  __syncthreads();   // g8 complete; the RGB copy in `scratch` is dead from here on

  // ---- pass 3: horizontal resample g8[H][W] -> tmp[H][OW]   (4 adjacent outputs per thread, one dword store)
  const uint8_t* hsrc = g8;
  int cur_w = a.W;
  if (a.W != a.OW) {
    const int groups = a.OW / 4;
    for (int i = tid; i < a.H * groups; i += FE_THREADS) {
      const int y = i / groups, x0 = (i - y * groups) * 4;
      uint32_t packed = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int xx = x0 + q, xmin = t_hb[xx * 2], n = t_hb[xx * 2 + 1];
        int32_t acc = 1 << (FE_PRECISION_BITS - 1);
        for (int k = 0; k < n; ++k) acc += (int32_t)g8[y * a.W + xmin + k] * t_hk[xx * a.hks + k];
        packed |= fe_clip8(acc) << (8 * q);
      }
      reinterpret_cast<uint32_t*>(scratch)[i] = packed;
    }
    __syncthreads();
    hsrc = scratch;
    cur_w = a.OW;
  }

  // ---- pass 4: vertical resample -> plane[OH][OW]; push into the agent's queue
  const int agent = a.agents ? a.agents[f] : -1;
  const bool clear = a.reset && a.reset[f];
  uint32_t* stack = agent >= 0 ? a.stacks + (size_t)agent * nout : nullptr;
  uint8_t* plane = a.planes ? a.planes + (size_t)f * nout : nullptr;
  const int groups = a.OW / 4;
  for (int i = tid; i < a.OH * groups; i += FE_THREADS) {
    const int yy = i / groups, x0 = (i - yy * groups) * 4;
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
    const int p0 = yy * a.OW + x0;
    if (plane) *reinterpret_cast<uint32_t*>(plane + p0) = v4[0] | (v4[1] << 8) | (v4[2] << 16) | (v4[3] << 24);
    if (stack) {
      uint4 s = clear ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(stack + p0);
      s.x = (s.x >> 8) | (v4[0] << 24);
      s.y = (s.y >> 8) | (v4[1] << 24);
      s.z = (s.z >> 8) | (v4[2] << 24);
      s.w = (s.w >> 8) | (v4[3] << 24);
      *reinterpret_cast<uint4*>(stack + p0) = s;
    }
  }
}

// bytes of dynamic LDS the kernel needs for a frame geometry
inline size_t frontend_lds_bytes(int H, int W, int C, int OH, int OW, int hks, int vks, bool cache) {
  const size_t tab = ((size_t)(OW * (2 + hks) + OH * (2 + vks)) * 4 + 15) & ~(size_t)15;
  const size_t g8 = ((size_t)H * W + 15) & ~(size_t)15;
  size_t scratch = (size_t)H * OW;
  if (cache && (size_t)H * W * C > scratch) scratch = (size_t)H * W * C;
  return tab + g8 + ((scratch + 15) & ~(size_t)15);
}

}  // namespace ga3c
