// ga3c_kernels.hpp -- hand-written gfx950 (CDNA4) kernels of the NetworkVP hot path.
//
// All contractions run on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains; the 1e-4 parity bar of
// BASELINE.json rules out bf16).  One wave owns one or more 16x16 output tiles.  Operand
// fragments follow the builtin's lane map (lane l: A[row l&15][k l>>4], B[k l>>4][col l&15],
// D[row 4*(l>>4)+reg][col l&15]).  The contraction index is consumed in a permuted order
// -- step s, lane group g, MFMA t  <->  k = 16 s + 4 g + t -- so that a lane's four k values
// of one step are contiguous in memory and arrive as ONE 16-byte load wherever the operand
// is k-contiguous.  Both operands of a product use the same permutation, so the sum is
// unchanged up to f32 summation order.
//
// Math restated from the reference graph (paths under /root/reference/ga3c):
//   conv + bias + ReLU   NetworkVP.py:212-228, wired as NetworkDNav.py:81-82 (SAME padding)
//   flatten + dense      NetworkDNav.py:86-90, :256-269
//   heads, softmax       NetworkVP_discrate.py:60,63,73-74
//   loss                 NetworkVP_discrate.py:61,64-85
//   RMSProp / clipping   NetworkVP_discrate.py:99-105,120-123,130
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ga3c {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IMG = 84, CIN = 4, XS = IMG * IMG * CIN;   // 28224 floats per state
constexpr int O1 = 21, C1 = 16, P1 = O1 * O1, N1S = P1 * C1;   // conv1 out: 441 px, 7056 floats
constexpr int O2 = 11, C2 = 32, P2 = O2 * O2, FLAT = P2 * C2;  // conv2 out: 121 px, 3872 floats
constexpr int HID = 256;
constexpr int KSTEPS_DENSE = FLAT / 16;                  // 242 permuted k-steps
constexpr int MAX_ACTIONS = 64;

// parameter arena offsets (floats), TensorFlow variable order
constexpr int64_t OFF_W1 = 0, OFF_B1 = OFF_W1 + 256 * 16, OFF_W2 = OFF_B1 + 16, OFF_B2 = OFF_W2 + 256 * 32,
                  OFF_WD = OFF_B2 + 32, OFF_BD = OFF_WD + (int64_t)FLAT * HID, OFF_WV = OFF_BD + HID,
                  OFF_BV = OFF_WV + HID, OFF_WP = OFF_BV + 1;
__host__ __device__ inline int64_t off_bp(int A) { return OFF_WP + (int64_t)HID * A; }
__host__ __device__ inline int64_t arena_floats(int A) { return off_bp(A) + A; }

// In-kernel time stamps for the DIAGNOSTIC build of tools/kprobe.hip only (-DGA3C_STAMPS): the product library never
// defines it, so no stamp executes there.  Stamps leave the kernel through a buffer of their own.
#ifdef GA3C_STAMPS
__device__ unsigned long long* ga3c_stamp_buf = nullptr;     // [workgroup][16 waves][16 stamps]
__device__ __forceinline__ void stamp_(int k) {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
#ifdef GA3C_STAMPS_REALTIME                                   // the 100 MHz counter all XCDs share (s_memtime is per XCD)
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
#else
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
#endif
  __builtin_amdgcn_sched_barrier(0);
  if ((threadIdx.x & 63) == 0 && ga3c_stamp_buf) ga3c_stamp_buf[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + k] = t;
}
#define GA3C_STAMP(k) stamp_(k)
#else
#define GA3C_STAMP(k)
#endif

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
// Pins a value in a register at this point of the program: an operand fetched from LDS ahead of time stays fetched ahead
// of time (hipcc otherwise sinks the read down to its use, re-uses one register for all of them and waits lgkmcnt(0) in
// front of every MFMA pair).
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// RMSProp applied by the kernel that PRODUCES a gradient element (single-GPU train steps without clipping): the optimizer's
// own launch, and its pass over the 4 MB dense1/w gradient, disappear.  Same arithmetic as rmsprop_one below
// (TF-1.x ApplyRMSProp: ms += (g*g - ms)*(1-rho); mom = mom*mu + g*lr/sqrt(eps+ms); theta_out = theta_in - mom).
struct FusedUpd {
  const float* tin; float* tout; float* ms; float* mom; float* pk;   // pk: packed copies (dense1/w, conv12/w) of tout
  float lr, omr, mu, eps; int on;
  int defer_wd;   // dense1/w is NOT stepped by the kernel that completes its gradient (dense1_bwd_tile) but by a later one
                  // (conv_bwd; conv2_dx_wd beyond 128 rows), beside that kernel's MFMA work: see wd_step_load / wd_step_apply
};
__device__ __forceinline__ float fused_rmsprop(const FusedUpd& u, int64_t i, float g) {
  float m = u.ms[i];
  m += (g * g - m) * u.omr;
  u.ms[i] = m;
  float step = (g * u.lr) / sqrtf(u.eps + m);
  if (u.mu != 0.f) {
    step = u.mom[i] * u.mu + step;
    u.mom[i] = step;
  }
  const float tn = u.tin[i] - step;
  u.tout[i] = tn;
  return tn;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------ state input: f32 NHWC or the uint8 frames
// The kernels that read the states take them either as the f32 tensor the reference feeds (NetworkVP.py:252) or as
// the uint8 frames they were made from (Environment.py:59-60), converting `k/128 - 1` while staging (exact in f32),
// so uint8 batches never exist as f32 in HBM.
__device__ __forceinline__ f32x4 px_from_u8(unsigned k) {   // the four channels of a pixel, k / 128 - 1 each
  return (f32x4){(float)(k & 255u) * 0.0078125f - 1.0f, (float)((k >> 8) & 255u) * 0.0078125f - 1.0f,
                 (float)((k >> 16) & 255u) * 0.0078125f - 1.0f, (float)(k >> 24) * 0.0078125f - 1.0f};
}
template <bool U8>
__device__ __forceinline__ f32x4 load_px(const void* __restrict__ x, size_t sample, int pix) {
  if (U8) return px_from_u8(reinterpret_cast<const unsigned*>(x)[sample * (XS / 4) + pix]);
  return ld4(reinterpret_cast<const float*>(x) + sample * XS + (size_t)pix * 4);
}

// ------------------------------------------------------------------ input conversion
// uint8 frames -> f32 `k/128 - 1` (Environment.py:60).  Exact in f32: k/128 is exact and so is the subtraction.
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                        int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const uchar4 k = reinterpret_cast<const uchar4*>(in)[i];
    f32x4 o = {(float)k.x * 0.0078125f - 1.0f, (float)k.y * 0.0078125f - 1.0f, (float)k.z * 0.0078125f - 1.0f,
               (float)k.w * 0.0078125f - 1.0f};
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// Rows of a batch gathered out of the registered host segment (the shm transport), over PCIe, as they are: row b =
// CHUNKS 16-byte chunks found at host_base + offsets[b] (28,224 uint8 frames: CHUNKS = 1764; 28,224 f32: 7056), written
// densely to dst.  uint8 frames stay uint8: the conv kernels convert while staging.
//
// Work item = (row, piece of 256 chunks): the row's offset is uniform over the workgroup (one scalar load out of the
// pinned offset array instead of a PCIe round trip per thread in front of every data load), four items are in flight per
// thread, and the grid is NARROW (ga3c_net::gather_max_blocks workgroups of 4 waves).  The kernel waits on the bus, not on the chip:
// launched as one chunk per thread it was 909 workgroups at 132 rows -- every wave slot of every CU held by waves waiting
// for PCIe -- and the train step of the OTHER trainer thread, whose kernels are one 16-wave workgroup per CU, could not get
// resident beside it: the staging of batch n+1 and step n ran one after the other although they sit on two streams.
// Narrower still is better for the step it runs beside: the reads in flight also sit in the memory system's queues for
// the length of a PCIe round trip, and the latency-bound kernels of the step (`heads`: 5 -> 37-63 us in the trace of
// profiles/README.md) wait behind them; 32 workgroups x 4 x 4 KB in flight still saturate the bus (72 us for 3.6 MB),
// 16 do not.  Two trainer threads, 128 rows, us per train call at 8 / 16 / 32 / 64 / 128 / 256 workgroups:
// 200 / 129 / 116 / 123 / 134 / 143.
inline int gather_blocks(int B, int chunks, int max_blocks) {
  const int items = B * ((chunks + 255) / 256);
  const int blocks = (items + 3) / 4;
  return blocks < max_blocks ? (blocks < 1 ? 1 : blocks) : max_blocks;
}

// small arrays that ride along with a gather (a batch's returns and one-hot actions, out of the pinned staging array):
// copied by the last workgroup, so that staging a batch is ONE launch instead of a gather and two copy kernels
struct SmallCopy { const float* src0; float* dst0; int n0; const float* src1; float* dst1; int n1; };
// by one workgroup of 256 threads; eight loads per thread are requested before the first is stored: the sources lie in
// pinned host memory, and a loop of load -> store paid a PCIe round trip per trip (four for a batch's one-hot actions)
__device__ __forceinline__ void small_copy(const SmallCopy& sc) {
  for (int pass = 0; pass < 2; ++pass) {
    const float* __restrict__ src = pass ? sc.src1 : sc.src0;
    float* __restrict__ dst = pass ? sc.dst1 : sc.dst0;
    const int n = pass ? sc.n1 : sc.n0;
    for (int i0 = 0; i0 < n; i0 += 2048) {
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + 256 * k + (int)threadIdx.x;
        t[k] = i < n ? src[i] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = i0 + 256 * k + (int)threadIdx.x;
        if (i < n) dst[i] = t[k];
      }
    }
  }
}

// ... and the rows' offsets travelling with the launch (n > 0) instead of being read out of the pinned array: every work
// item's data loads waited for a scalar load over PCIe in front of them -- a round trip per trip of the loop
struct RowOffsets { int64_t off[192]; int n; };

template <int CHUNKS>
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* __restrict__ host_base,
                                                          const int64_t* __restrict__ offsets, uint4* __restrict__ dst, int B,
                                                          SmallCopy sc, const RowOffsets ro) {
  constexpr int PIECES = (CHUNKS + 255) / 256;
  const int items = B * PIECES;
  const int tid = threadIdx.x;
  if (blockIdx.x == gridDim.x - 1) small_copy(sc);
  for (int it0 = blockIdx.x * 4; it0 < items; it0 += gridDim.x * 4) {
    uint4 v[4];
    bool live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = it0 + u;
      const int b = it / PIECES, c = (it - b * PIECES) * 256 + tid;
      live[u] = it < items && c < CHUNKS;
      if (live[u]) v[u] = *reinterpret_cast<const uint4*>(host_base + (ro.n ? ro.off[b < 192 ? b : 0] : offsets[b]) + (size_t)c * 16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = it0 + u;
      const int b = it / PIECES, c = (it - b * PIECES) * 256 + tid;
      if (live[u]) dst[(size_t)b * CHUNKS + c] = v[u];
    }
  }
}

// Rows of a batch copied out of the state cache (HBM to HBM): workgroup = (256-chunk piece, row); the row's offset comes out
// of the kernel arguments.  (gather_rows_kernel, built to wait on the bus with few waves, needed 50-60 us for these 3.7 MB.)
template <int CHUNKS>
__global__ __launch_bounds__(256) void copy_rows_kernel(const uint8_t* __restrict__ base, const int64_t* __restrict__ offsets,
                                                        uint4* __restrict__ dst, int B, SmallCopy sc, const RowOffsets ro) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && b == 0) small_copy(sc);
  // (batches beyond the 192 offsets that fit the arguments read them out of the pinned array: ro.n == 0)
  if (b < B && c < CHUNKS) dst[(size_t)b * CHUNKS + c] = *reinterpret_cast<const uint4*>(base + (ro.n ? ro.off[b < 192 ? b : 0] : offsets[b]) + (size_t)c * 16);
}

// ... and the other way: the dense uint8 rows of a batch (gathered for a prediction step beyond the fused conv stack's 128 rows)
// filed into the state cache, row b at base + dst_off[b] (offsets in a pinned array)
template <int CHUNKS>
__global__ __launch_bounds__(256) void file_rows_kernel(const uint4* __restrict__ src, uint8_t* __restrict__ base,
                                                        const int64_t* __restrict__ dst_off, int B) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (b < B && c < CHUNKS) *reinterpret_cast<uint4*>(base + dst_off[b] + (size_t)c * 16) = src[(size_t)b * CHUNKS + c];
}

// ------------------------------------------------------------------ conv1 forward
// n1[m][o] = relu(b1[o] + sum_k patch(m)[k] W1[k][o]),  m = (b*21+i)*21+j, k = (u*8+v)*4+c.
// Implicit GEMM M = B*441, K = 256, N = 16.  im2col re-reads every input pixel 4x; served from L2 that
// re-read was the kernel's bottleneck, so a workgroup = (sample, band of 3 output rows) first copies
// the band's 16 input rows, zero padding included, into LDS (every byte fetched once, coalesced) and
// the waves then gather patch rows with 16-byte LDS reads.  W1 is staged through LDS once per block in
// MFMA fragment order and kept in 64 VGPRs per lane.
constexpr int C1_HB = 3;                    // output rows per band (21 = 7 bands)
constexpr int C1_RIN = 4 * C1_HB + 4;       // padded input rows per band
constexpr int C1_PW = 88;                   // padded input width (2 left, 2 right)
constexpr int C1_LDS_FLOATS = C1_RIN * C1_PW * 4 + 64 * 64;

template <bool U8>
__global__ __launch_bounds__(256, 3) void conv1_fwd_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ n1,
                                                           int B) {
  __shared__ __attribute__((aligned(16))) float lds[C1_LDS_FLOATS];
  float* img = lds;                         // [C1_RIN][88] pixels x 4 channels
  float* wl = lds + C1_RIN * C1_PW * 4;     // [j = s*4+t][lane = g*16+r] = W1[16s+4g+t][r]
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / 7, band = blockIdx.x - b * 7;
  if (b >= B) return;                       // block-uniform guard: the grid is B * 7
  const int y_base = 4 * C1_HB * band - 2;  // image row of padded band row 0
  // band rows -> LDS (zero fill outside the image).  uint8 states as 16 x 21 pieces of 16 bytes = four pixels (336 wide loads
  // instead of 1,408 dword loads), converted on the way into LDS; the padding columns no piece covers are zeroed apart
  f32x4 stage[U8 ? 1 : 6];
  uint4 raw[2];
  constexpr int U8_PIECES = C1_RIN * (IMG / 4);
  if constexpr (U8) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = threadIdx.x + 256 * i;
      const int row = t / (IMG / 4), grp = t - row * (IMG / 4);
      const int yy = y_base + row;
      const bool ok = t < U8_PIECES && (unsigned)yy < (unsigned)IMG;
      raw[i] = ok ? *reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(x) + (size_t)b * XS + (size_t)(yy * IMG + 4 * grp) * 4)
                  : make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);     // 128 / 128 - 1 = 0
    }
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int row = idx / C1_PW, col = idx - row * C1_PW;
      const int yy = y_base + row, xx = col - 2;
      const bool ok = idx < C1_RIN * C1_PW && (unsigned)yy < (unsigned)IMG && (unsigned)xx < (unsigned)IMG;
      stage[i] = ok ? load_px<U8>(x, b, yy * IMG + xx) : zero4();
    }
  }
  f32x4 wstage[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wstage[i] = ld4(w + 4 * (threadIdx.x + 256 * i));
  if constexpr (U8) {
    if (threadIdx.x < 4 * C1_RIN) {
      const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
      *reinterpret_cast<f32x4*>(&img[(row * C1_PW + (c < 2 ? c : IMG + c)) * 4]) = zero4();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = threadIdx.x + 256 * i;
      if (t < U8_PIECES) {
        const int row = t / (IMG / 4), grp = t - row * (IMG / 4);
        float* dst = img + (row * C1_PW + 2 + 4 * grp) * 4;
        *reinterpret_cast<f32x4*>(dst) = px_from_u8(raw[i].x);
        *reinterpret_cast<f32x4*>(dst + 4) = px_from_u8(raw[i].y);
        *reinterpret_cast<f32x4*>(dst + 8) = px_from_u8(raw[i].z);
        *reinterpret_cast<f32x4*>(dst + 12) = px_from_u8(raw[i].w);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < C1_RIN * C1_PW) *reinterpret_cast<f32x4*>(&img[idx * 4]) = stage[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx4 = threadIdx.x + 256 * i;          // float4 index into W1[256][16]
    const int k = idx4 >> 2, n = (idx4 & 3) * 4;
    const int j = (k >> 4) * 4 + (k & 3), gg = (k >> 2) & 3;
    *reinterpret_cast<f32x4*>(&wl[j * 64 + gg * 16 + n]) = wstage[i];
  }
  __syncthreads();
  float wr[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) wr[j] = wl[j * 64 + lane];
  const float bv = bias[r];
  // 63 pixels per band = 4 tiles of 16: one per wave
  const int ml = wv * 16 + r;
  const bool valid = ml < C1_HB * O1;
  const int mm = valid ? ml : 0;
  const int il = mm / O1, j = mm - il * O1;
  const float* base = img + ((4 * il) * C1_PW + 4 * j + g) * 4;
  f32x4 a[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) a[s] = ld4(base + ((s >> 1) * C1_PW + (s & 1) * 4) * 4);
  f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
  for (int s = 0; s < 16; s += 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = mfma(a[s][t], wr[s * 4 + t], acc0);
      acc1 = mfma(a[s + 1][t], wr[(s + 1) * 4 + t], acc1);
    }
  }
  float* out = n1 + ((size_t)b * P1 + band * C1_HB * O1) * C1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int mr = wv * 16 + 4 * g + q;
    if (mr < C1_HB * O1) out[(size_t)mr * C1 + r] = fmaxf(acc0[q] + acc1[q] + bv, 0.f);
  }
}

// ------------------------------------------------------------------ conv2 forward
// n2[m][o] = relu(b2[o] + sum_k patch(m)[k] W2[k][o]), m = (b*11+i)*11+j, k = (u*4+v)*16+c.
// M = B*121, K = 256, N = 32.  Workgroup = (sample, 16-column half of the output): the sample's n1
// (28 KB, zero-padded to 24x24 pixels) and the half of W2 it needs go through LDS; 8 tiles of 16
// output pixels, two per wave.
constexpr int C2_PW = 24;                   // padded n1 width/height (1 before, 2 after)
constexpr int C2_LDS_FLOATS = C2_PW * C2_PW * C1 + 64 * 64;

// Q: a workgroup takes FOUR of the eight tiles (grid B * 4).  A unit of eight tiles is 128 MFMAs per SIMD: from 129 rows on some
// CUs hold two units and the kernel takes the second one's MFMA time on top (5.5 -> 7.4 us at 132 rows, flat to 256); with units
// of four tiles the heaviest CU holds three (1.5 units) at the price of staging the image twice as often.
template <bool Q = false>
__global__ __launch_bounds__(256, 2) void conv2_fwd_kernel(const float* __restrict__ n1, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ n2,
                                                           int B) {
  __shared__ __attribute__((aligned(16))) float lds[C2_LDS_FLOATS];
  float* img = lds;                          // [24][24] pixels x 16 channels
  float* wl = lds + C2_PW * C2_PW * C1;      // [j][lane] = W2[16s+4g+t][half*16 + r]
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  const int b = Q ? blockIdx.x >> 2 : blockIdx.x >> 1, half = blockIdx.x & 1;
  const int rep0 = Q ? (blockIdx.x >> 1) & 1 : 0;
  if (b >= B) return;                       // block-uniform guard: the grid is B * 2 (B * 4)
  const float* nb = n1 + (size_t)b * N1S;
  // padded image: 24*24*4 = 2304 float4 -> 9 per thread
  f32x4 stage[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int idx = threadIdx.x + 256 * i;           // float4 index: pixel = idx>>2, channel quad = idx&3
    const int px = idx >> 2, row = px / C2_PW, col = px - row * C2_PW;
    const int yy = row - 1, xx = col - 1;
    const bool ok = (unsigned)yy < (unsigned)O1 && (unsigned)xx < (unsigned)O1;
    stage[i] = ok ? ld4(nb + (yy * O1 + xx) * C1 + (idx & 3) * 4) : zero4();
  }
  f32x4 wstage[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q4 = threadIdx.x + 256 * i;            // float4 index into this half: k = q4>>2, n = (q4&3)*4
    wstage[i] = ld4(w + (q4 >> 2) * C2 + half * 16 + (q4 & 3) * 4);
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) *reinterpret_cast<f32x4*>(&img[(threadIdx.x + 256 * i) * 4]) = stage[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q4 = threadIdx.x + 256 * i;
    const int k = q4 >> 2, n = (q4 & 3) * 4;
    const int j = (k >> 4) * 4 + (k & 3), gg = (k >> 2) & 3;
    *reinterpret_cast<f32x4*>(&wl[j * 64 + gg * 16 + n]) = wstage[i];
  }
  __syncthreads();
  float wr[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) wr[j] = wl[j * 64 + lane];
  const float bv = bias[half * 16 + r];
#pragma unroll
  for (int it = 0; it < (Q ? 1 : 2); ++it) {
    const int rep = Q ? rep0 : it;
    const int tile = wv + 4 * rep;                   // 8 tiles cover 121 pixels
    const int ml = tile * 16 + r;
    const bool valid = ml < P2;
    const int mm = valid ? ml : 0;
    const int i = mm / O2, j = mm - i * O2;
    const float* base = img + ((2 * i) * C2_PW + 2 * j) * C1 + 4 * g;
    f32x4 a[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = ld4(base + ((s >> 2) * C2_PW + (s & 3)) * C1);
    f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = mfma(a[s][t], wr[s * 4 + t], acc0);
        acc1 = mfma(a[s + 1][t], wr[(s + 1) * 4 + t], acc1);
      }
    }
    float* out = n2 + (size_t)b * FLAT + half * 16 + r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mr = tile * 16 + 4 * g + q;
      if (mr < P2) out[(size_t)mr * C2] = fmaxf(acc0[q] + acc1[q] + bv, 0.f);
    }
  }
}

// ------------------------------------------------------------------ conv1 + conv2 forward in one launch
// Workgroup (16 waves) = (sample, half of the conv2 output).  The x rows the half needs (56 or 52 padded rows,
// <= 77 KB) and both filter banks go to LDS once; conv1 runs tile by tile and leaves its ReLU output in an LDS image
// of n1 laid out for conv2 (zero border included), conv2 then reads its patches from that image.  n1 never travels
// to HBM in prediction (TRAIN also stores it: the backward pass needs it).
// The 121 conv2 pixels are cut by PIXEL, 60 | 61, not by row: each half then has exactly 4 conv2 tiles (8 items of 64
// MFMAs, two per SIMD) and at most 16 conv1 tiles (one per wave, four per SIMD) -- the row cut 55 | 66 left the lower
// half with 5 tiles, i.e. three items on two of its SIMDs, and that half set the kernel's length.  The n1 pixels a half
// needs are then ragged: upper = rows 0..10 whole + rows 11,12 columns 0..10 (253 px); lower = rows 9,10 columns 9..20 +
// rows 11..20 whole (234 px); 46 of the 441 n1 pixels are computed by both halves (identical values).
// Saves one launch and the 3.6 MB write + 7.2 MB read of n1 per 128 states against conv1_fwd + conv2_fwd.
constexpr int CS_XROWS = 56;                                  // upper half: x rows -2..53; lower half uses 52 (34..85)
constexpr int CS_X_FLOATS = CS_XROWS * C1_PW * 4;             // 19712
constexpr int CS_N1ROWS = 14;                                 // upper: n1 rows -1..12; lower: 9..22
constexpr int CS_N1_FLOATS = CS_N1ROWS * C2_PW * C1;          // 5376
constexpr int CS_LDS_FLOATS = CS_X_FLOATS + CS_N1_FLOATS + 64 * 64 + 2 * 64 * 64 + 256;   // 37632 floats = 150,528 B
constexpr int CS_C2CUT = 60;                                  // conv2 pixels [0,60) | [60,121)

// (sample, half) of workgroup `id` of a 2 * B grid.  The two halves of a sample are EIGHT workgroups apart: blocks are dealt
// round-robin over the 8 XCDs, so both halves run on the same XCD at the same time, and the 24 x rows (conv_bwd: n1 / dn2
// rows) both of them need come out of HBM once and out of that XCD's L2 the second time (adjacent blocks, the round-2
// mapping, put the halves on different XCDs: 20.4 MB fetched for 16.5 MB of states).  A placement hint only: any placement
// computes the same result.  The last, partial group of samples (B % 8 of them) keeps its halves B % 8 apart.
__device__ __forceinline__ void cs_sample_half(int id, int B, int& b, int& h) {
  const int full = B >> 3, G = id >> 4;
  if (G < full) {
    b = 8 * G + (id & 7);
    h = (id >> 3) & 1;
  } else {
    const int m = B - 8 * full, rr = id - 16 * full;      // m >= 1: id < 2 * B
    h = rr >= m ? 1 : 0;
    b = 8 * full + rr - h * m;
  }
}

// the m-th n1 pixel of a half's ragged list -> (row, col)
__device__ __forceinline__ void cs_n1_pixel(int h, int m, int& row, int& col) {
  if (h == 0) {
    if (m < 11 * O1) { row = m / O1; col = m - row * O1; }
    else { const int e = m - 11 * O1; row = 11 + e / 11; col = e - (e / 11) * 11; }
  } else {
    if (m < 24) { row = 9 + m / 12; col = 9 + m - (m / 12) * 12; }
    else { const int e = m - 24; row = 11 + e / O1; col = e - (e / O1) * O1; }
  }
}

// The byte offsets of a scattered batch travelling with the launch (kernel arguments live in device memory: a scalar load
// out of HBM) instead of being read out of the pinned host array the caller filled -- that read was a PCIe round trip of its
// own in front of the staging loads that depend on it.  n = 0: read src_off (hipGraph replays: arguments are baked in).
// dst (n_dst > 0): the workgroups of sample b also store the uint8 state they stage at cache + dst[b] (the state cache:
// a train batch can then NAME the state instead of carrying it over the bus a second time; ga3c_net_state_cache_config).
struct SrcOffsets { int64_t off[128]; int64_t dst[128]; int n; int n_dst; };

// w1f / w2f: the filter banks in forward fragment order (theta_pk + PK_W1F / PK_W2F)
template <bool TRAIN, bool U8>
__global__ __launch_bounds__(1024) void conv_stack_fwd_kernel(const void* __restrict__ x, const float* __restrict__ w1f,
                                                             const float* __restrict__ b1, const float* __restrict__ w2f,
                                                             const float* __restrict__ b2, float* __restrict__ n1,
                                                             float* __restrict__ n2, int B,
                                                             const int64_t* __restrict__ src_off, const SrcOffsets so,
                                                             uint8_t* __restrict__ cache) {
  extern __shared__ __attribute__((aligned(16))) float cs_lds[];
  float* img = cs_lds;
  float* n1l = cs_lds + CS_X_FLOATS;
  float* wl1 = n1l + CS_N1_FLOATS;
  float* wl2 = wl1 + 64 * 64;
  int* pxmap = reinterpret_cast<int*>(wl2 + 2 * 64 * 64);    // ragged n1 pixel list: m -> (row << 8) | col
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  if ((int)blockIdx.x >= 2 * B) return;                      // block-uniform guard: the grid is B * 2
  int b, h;
  cs_sample_half(blockIdx.x, B, b, h);
  // src_off: the states are not a dense batch but lie src_off[b] bytes behind x (transport slots in registered host
  // memory, or the device-side frame queues): the intake gather happens here, in the staging loads
  const void* xs = src_off ? static_cast<const void*>(static_cast<const char*>(x) + (so.n ? so.off[b] : src_off[b])) : x;
  const size_t sb = src_off ? 0 : (size_t)b;
  const int q0 = h ? CS_C2CUT : 0, c2npix = h ? P2 - CS_C2CUT : CS_C2CUT;     // conv2 pixels of this half
  const int c2r0 = h ? 5 : 0;                                // first conv2 row it touches
  const int n1r0 = h ? 9 : 0;                                // first n1 row it needs
  const int n1npix = h ? 24 + 10 * O1 : 11 * O1 + 22;        // ragged n1 pixel list: 234 | 253
  const int n1own0 = h ? 11 : 0, n1own1 = h ? O1 : 11;       // TRAIN: n1 rows this half stores (each row exactly once)
  const int n1org = 2 * c2r0 - 1;                            // n1 row held by LDS image row 0
  const int xr0 = 4 * n1r0 - 2, xnr = h ? 52 : 56;           // x rows it needs
  // ---- stage x rows (zero padded) and W1, clear the n1 image.  Every global load is issued before anything waits, so
  // the block pays one memory round trip.  f32 states go global -> LDS directly (LDS-DMA: no VGPR pass, no ds_write; the
  // image is linear in idx, so a wave's 64 pixels are 1 KB contiguous = wave base + lane * 16, and the padding cells,
  // which no DMA lane writes, are zeroed by ordinary stores); uint8 frames are converted in registers on the way.
  // W2 is not needed before conv2: its loads are in flight too, but it goes to LDS behind this wave's conv1 tile, under
  // the other waves' MFMAs.
  const int npx = xnr * C1_PW;                               // <= 4928 pixels -> at most 5 per thread
  GA3C_STAMP(0);
  f32x4 sx[5];
  // W1 (16 KB, needed first) then, behind the x rows, W2 (32 KB): packed, so they go global -> LDS as they lie
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w1f + 4 * threadIdx.x),
                                   (__attribute__((address_space(3))) void*)(wl1 + 4 * (threadIdx.x - lane)), 16, 0, 0);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int idx = threadIdx.x + 1024 * i;
    const int row = idx / C1_PW, col = idx - row * C1_PW;
    const int yy = xr0 + row, xx = col - 2;
    const bool ok = idx < npx && (unsigned)yy < (unsigned)IMG && (unsigned)xx < (unsigned)IMG;
    if (!U8) {
      float* dst = img + (size_t)(idx - lane) * 4;           // wave-uniform
      if (ok)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(xs) + sb * XS + (size_t)(yy * IMG + xx) * 4),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      else if (idx < npx)
        *reinterpret_cast<f32x4*>(&img[idx * 4]) = zero4();
    } else {
      const unsigned k = ok ? reinterpret_cast<const unsigned*>(xs)[sb * (XS / 4) + yy * IMG + xx] : 0u;
      sx[i] = ok ? px_from_u8(k) : zero4();
      // the state cache keeps the bytes as they came: each half stores the image rows it owns (0..41 | 42..83), once
      if (so.n_dst && ok && (h ? yy >= IMG / 2 : yy < IMG / 2)) reinterpret_cast<unsigned*>(cache + so.dst[b])[yy * IMG + xx] = k;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w2f + 4 * (threadIdx.x + 1024 * i)),
                                     (__attribute__((address_space(3))) void*)(wl2 + 4 * (threadIdx.x + 1024 * i - lane)), 16, 0, 0);
  // nothing in the next two statements depends on the loads: they run while the loads are in flight
  for (int idx4 = threadIdx.x; idx4 < CS_N1_FLOATS / 4; idx4 += 1024) *reinterpret_cast<f32x4*>(&n1l[idx4 * 4]) = zero4();
  if (threadIdx.x < 256) {                                   // the divisions of the ragged map, once per pixel
    int row, col;
    cs_n1_pixel(h, threadIdx.x < n1npix ? threadIdx.x : 0, row, col);
    pxmap[threadIdx.x] = (row << 8) | col;
  }
  GA3C_STAMP(1);
  if (U8) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = threadIdx.x + 1024 * i;
      if (idx < npx) *reinterpret_cast<f32x4*>(&img[idx * 4]) = sx[i];
    }
  }
  GA3C_STAMP(2);
  GA3C_STAMP(3);
  __syncthreads();                                           // (waits for the DMA too: vmcnt(0) precedes the barrier)
  GA3C_STAMP(4);
  // ---- conv1 over the half's n1 pixels, result into the LDS image (and to HBM when training)
  {
    const float bv = b1[r];
    const int ntile = (n1npix + 15) >> 4;                    // 15 | 16: one tile per wave
    for (int tile = wv; tile < ntile; tile += 16) {
      const int e = pxmap[tile * 16 + r];
      const int row = e >> 8, col = e & 255;
      const float* base = img + ((4 * (row - n1r0)) * C1_PW + 4 * col + g) * 4;
      // per pair of k-steps: 2 patch reads (16 B) + 8 filter fragments, fetched and pinned one pair ahead of their 8 MFMAs
      f32x4 pa[2][2], pw[2][2];
      auto load_pair = [&](int s, f32x4 (&a)[2], f32x4 (&w2)[2]) {
        a[0] = ld4(base + ((s >> 1) * C1_PW + (s & 1) * 4) * 4);
        a[1] = ld4(base + (((s + 1) >> 1) * C1_PW + ((s + 1) & 1) * 4) * 4);
        w2[0] = ld4(wl1 + (s * 64 + lane) * 4);
        w2[1] = ld4(wl1 + ((s + 1) * 64 + lane) * 4);
        pin(a[0]); pin(a[1]); pin(w2[0]); pin(w2[1]);
      };
      f32x4 acc0 = zero4(), acc1 = zero4();
      load_pair(0, pa[0], pw[0]);
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        if (s + 2 < 16) load_pair(s + 2, pa[((s >> 1) + 1) & 1], pw[((s >> 1) + 1) & 1]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc0 = mfma(pa[(s >> 1) & 1][0][t], pw[(s >> 1) & 1][0][t], acc0);
          acc1 = mfma(pa[(s >> 1) & 1][1][t], pw[(s >> 1) & 1][1][t], acc1);
        }
      }
      const int4 e4 = *reinterpret_cast<const int4*>(&pxmap[tile * 16 + 4 * g]);
      const int em[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (tile * 16 + 4 * g + q < n1npix) {
          const int ir = em[q] >> 8, jr = em[q] & 255;
          const float val = fmaxf(acc0[q] + acc1[q] + bv, 0.f);
          n1l[((ir - n1org) * C2_PW + jr + 1) * C1 + r] = val;
        }
      }
    }
  }
  GA3C_STAMP(5);
  __syncthreads();
  GA3C_STAMP(6);
  if (TRAIN) {   // the rows of n1 this half owns (each row exactly once), out of the LDS image as whole 16-byte pieces; under conv2
    const int row = threadIdx.x / (O1 * C1 / 4), c4 = threadIdx.x - row * (O1 * C1 / 4);
    const int ir = n1own0 + row;
    if (ir < n1own1)
      *reinterpret_cast<f32x4*>(n1 + ((size_t)b * P1 + ir * O1) * C1 + 4 * c4) =
          *reinterpret_cast<const f32x4*>(n1l + ((ir - n1org) * C2_PW + 1) * C1 + 4 * c4);
  }
  // ---- conv2 from the LDS image: items = (16-pixel tile, 16-column half), 8 per workgroup
  {
    const int nitem = 2 * ((c2npix + 15) >> 4);
    for (int item = wv; item < nitem; item += 16) {
      const int tile = item >> 1, hh = item & 1;
      const float* wf = wl2 + hh * 4096;
      const int ml = tile * 16 + r;
      const int qq = q0 + (ml < c2npix ? ml : 0);
      const int i2 = qq / O2, j2 = qq - i2 * O2;
      const float* base = n1l + ((2 * (i2 - c2r0)) * C2_PW + 2 * j2) * C1 + 4 * g;
      f32x4 pa[2][2], pw[2][2];
      auto load_pair = [&](int s, f32x4 (&a)[2], f32x4 (&w2)[2]) {
        a[0] = ld4(base + ((s >> 2) * C2_PW + (s & 3)) * C1);
        a[1] = ld4(base + (((s + 1) >> 2) * C2_PW + ((s + 1) & 3)) * C1);
        w2[0] = ld4(wf + (s * 64 + lane) * 4);
        w2[1] = ld4(wf + ((s + 1) * 64 + lane) * 4);
        pin(a[0]); pin(a[1]); pin(w2[0]); pin(w2[1]);
      };
      f32x4 acc0 = zero4(), acc1 = zero4();
      load_pair(0, pa[0], pw[0]);
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        if (s + 2 < 16) load_pair(s + 2, pa[((s >> 1) + 1) & 1], pw[((s >> 1) + 1) & 1]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc0 = mfma(pa[(s >> 1) & 1][0][t], pw[(s >> 1) & 1][0][t], acc0);
          acc1 = mfma(pa[(s >> 1) & 1][1][t], pw[(s >> 1) & 1][1][t], acc1);
        }
      }
      const float bv = b2[hh * 16 + r];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mr = tile * 16 + 4 * g + q;
        if (mr < c2npix) n2[(size_t)b * FLAT + (size_t)(q0 + mr) * C2 + hh * 16 + r] = fmaxf(acc0[q] + acc1[q] + bv, 0.f);
      }
    }
  }
  GA3C_STAMP(7);
}

// ------------------------------------------------------------------ conv12/w packed for conv2_dx
// conv2_dx contracts dn2 with the 128 x 16 sub-matrix of W2 that belongs to an output-pixel parity class (py, px); its
// B fragments, in the order the MFMAs consume them, are a permutation of W2's 8192 elements: element (u, v, c, o) of
// W2[4,4,16,32] goes to class (1 - u%2, 1 - v%2), step s = (u/2)*4 + (v/2)*2 + o/16, MFMA t = o%4 of that step,
// lane = ((o%16)/4)*16 + c, stored [class][s][lane][t]: the four B fragments a lane needs for the four MFMAs of a step are
// ONE 16-byte LDS read (round 3; [class][s*4+t][lane] before: one 4-byte read per MFMA).  The packed copy lives behind
// dense1's packed copy (pk + FLAT*HID) and is kept current by rmsprop_kernel / slab_reduce's fused update /
// pack_w2dx_kernel, so a workgroup stages a class with ONE 16-byte load per thread.
constexpr int64_t PK_W2DX = (int64_t)FLAT * HID;             // offset of the packed conv12/w inside the pk buffer
// Behind it, the two conv filter banks in the order conv_stack_fwd's MFMAs consume them (round 3): element (k, n) of
// conv11/w[256][16] at [s = k/16][lane = ((k/4)%4)*16 + n][t = k%4], element (k, o) of conv12/w[256][32] at
// [column half o/16][s][lane = ((k/4)%4)*16 + o%16][t]: the four B fragments of a step are ONE 16-byte LDS read, and the
// banks are staged by LDS-DMA as they lie (no pass through registers, no scattered ds_write).
constexpr int64_t PK_W1F = PK_W2DX + 256 * 32;               // conv11/w, forward fragment order (4096 floats)
constexpr int64_t PK_W2F = PK_W1F + 256 * 16;                // conv12/w, forward fragment order (8192 floats)
constexpr int PK_FLOATS = FLAT * HID + 256 * 32 + 256 * 16 + 256 * 32;
__host__ __device__ inline int w1f_packed_index(int i) {    // i = k*16 + n
  const int n = i & 15, k = i >> 4;
  return (((k >> 4) * 64 + ((k >> 2) & 3) * 16 + n) << 2) | (k & 3);
}
__host__ __device__ inline int w2f_packed_index(int i) {    // i = k*32 + o
  const int o = i & 31, k = i >> 5;
  return (o >> 4) * 4096 + ((((k >> 4) * 64 + ((k >> 2) & 3) * 16 + (o & 15)) << 2) | (k & 3));
}
__host__ __device__ inline int w2dx_packed_index(int i) {   // i = index into W2[256][32] = ((u*4+v)*16 + c)*32 + o
  const int o = i & 31, k = i >> 5, c = k & 15, uv = k >> 4, u = uv >> 2, v = uv & 3;
  const int cls = (1 - (u & 1)) * 2 + (1 - (v & 1));
  const int s = ((u >> 1) << 2) | ((v >> 1) << 1) | (o >> 4);
  const int t = o & 3, ln = ((o & 15) >> 2) * 16 + c;
  return cls * 2048 + (s * 64 + ln) * 4 + t;
}
// the packed copies of a conv parameter at arena index i (conv11/w or conv12/w; no-op elsewhere)
__device__ __forceinline__ void store_conv_packs(float* pk, int64_t i, float v) {
  if (i >= OFF_W2 && i < OFF_B2) {
    pk[PK_W2DX + w2dx_packed_index((int)(i - OFF_W2))] = v;
    pk[PK_W2F + w2f_packed_index((int)(i - OFF_W2))] = v;
  } else if (i < OFF_B1) {
    pk[PK_W1F + w1f_packed_index((int)i)] = v;
  }
}
// theta -> all packed copies of the two conv filter banks (pk = base of the packed buffer); grid 48 x 256
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ theta, float* __restrict__ pk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 256 * 32) store_conv_packs(pk, OFF_W2 + i, theta[OFF_W2 + i]);
  else if (i < 256 * 32 + 256 * 16) store_conv_packs(pk, OFF_W1 + (i - 256 * 32), theta[OFF_W1 + (i - 256 * 32)]);
}

// ------------------------------------------------------------------ dense1 weight packing
// pk[(s*256 + n)*16 + kk] = Wd[16 s + kk][n]: the B-operand fragment order of dense1_fwd, so that a lane's
// four k values of a step are one 16-byte load and a wave instruction reads 1 KB contiguously.
// One block per 16-row group (both layouts keep a group in the same 16 KB span).
__global__ __launch_bounds__(256) void pack_wd_kernel(const float* __restrict__ wd, float* __restrict__ pk) {
  __shared__ float tile[16][HID + 1];
  const int s = blockIdx.x, n = threadIdx.x;
  const float* src = wd + (size_t)s * 16 * HID;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) tile[kk][n] = src[kk * HID + n];
  __syncthreads();
  float* dst = pk + ((size_t)s * HID + n) * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {tile[4 * q][n], tile[4 * q + 1][n], tile[4 * q + 2][n], tile[4 * q + 3][n]};
    *reinterpret_cast<f32x4*>(dst + 4 * q) = v;
  }
}

// ------------------------------------------------------------------ dense1 forward (split-K)
// part[ks][b][n] = sum_{k in slice ks} flat[b][k] Wd[k][n];  M = B, N = 256, K = 3872 = 242 steps of 16.
// Wave tile (16*MT rows) x 32 columns, operands as 16-byte loads (flat rows; packed Wd), next step's
// operands in flight during the current step's MFMAs.  1-D grid, XCD-aware: blocks are dealt round-robin
// to the 8 XCDs, so block id -> (xcd = id % 8, j = id / 8); all row blocks that read the same slice of Wd
// (same ks, same column half) get the same xcd and share that XCD's L2 copy of it.
template <int MT>
__global__ __launch_bounds__(256) void dense1_fwd_kernel(const float* __restrict__ flat, const float* __restrict__ pk,
                                                         float* __restrict__ part, int B, int ks_total,
                                                         int steps_per_slice) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int nrow = (B + 16 * MT - 1) / (16 * MT);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int rb = j % nrow, yz = (j / nrow) * 8 + xcd;      // yz = ks * 2 + column half
  if (yz >= ks_total * 2) return;
  const int m0 = rb * 16 * MT, ks = yz >> 1;
  const int n0 = ((yz & 1) * 4 + (threadIdx.x >> 6)) * 32;
  bool v[MT];
  const float* ap[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    v[mi] = m0 + 16 * mi + r < B;
    ap[mi] = flat + (size_t)(v[mi] ? m0 + 16 * mi + r : 0) * FLAT + 4 * g;
  }
  const float* w0p = pk + (size_t)(n0 + r) * 16 + 4 * g;
  const float* w1p = w0p + 16 * 16;
  // two accumulators per output tile (k steps t = 0, 2 | 1, 3), added at the end: the summation order of
  // dense1_fwd_tile_kernel, so that the two kernels give the same bits and the engine may pick either per step
  f32x4 acc[MT][2][2];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) acc[mi][0][0] = acc[mi][0][1] = acc[mi][1][0] = acc[mi][1][1] = zero4();
  // slice ks of ks_total: the 242 K steps are cut as evenly as integers allow (steps_per_slice < 0), so that the slice
  // count can be picked to fill whole rounds of workgroups on the 256 CUs rather than to divide 242
  const int s0 = steps_per_slice > 0 ? ks * steps_per_slice : (ks * KSTEPS_DENSE) / ks_total;
  const int s1 = steps_per_slice > 0 ? s0 + steps_per_slice : ((ks + 1) * KSTEPS_DENSE) / ks_total;
  // D steps of operands in flight (a ring of register slots, the loop unrolled by D so that slot numbers are compile-time):
  // a step is a dependent L2 round trip, and one step of prefetch left the wave waiting on most of it
  constexpr int D = MT == 1 ? 4 : 2;
  f32x4 a[D][MT], w0[D], w1[D];
  auto fetch = [&](int slot, int s) {
    if (s < s1) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) a[slot][mi] = v[mi] ? ld4(ap[mi] + 16 * s) : zero4();
      w0[slot] = ld4(w0p + (size_t)s * HID * 16);
      w1[slot] = ld4(w1p + (size_t)s * HID * 16);
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) fetch(d, s0 + d);
  for (int s = s0; s < s1; s += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (s + d < s1) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) {
            acc[mi][0][t & 1] = mfma(a[d][mi][t], w0[d][t], acc[mi][0][t & 1]);
            acc[mi][1][t & 1] = mfma(a[d][mi][t], w1[d][t], acc[mi][1][t & 1]);
          }
        fetch(d, s + d + D);
      }
    }
  }
  float* out = part + ((size_t)ks * B) * HID + n0 + r;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mr = m0 + mi * 16 + 4 * g + q;
      if (mr < B) {
        out[(size_t)mr * HID] = acc[mi][0][0][q] + acc[mi][0][1][q];
        out[(size_t)mr * HID + 16] = acc[mi][1][0][q] + acc[mi][1][1][q];
      }
    }
}
// ---- the same partial products, LDS-tiled: a workgroup (8 waves) = (16*MT rows, 128 columns, one K slice of <= 16 steps).
// Its whole working set -- the slice of the packed Wd (<= 128 KB, 1 KB per step and 16 columns = one LDS-DMA wave
// instruction) and the rows' flat slice (1 KB per row, padded to 260 floats) -- is requested up front and lands in LDS
// without a VGPR pass; the MFMAs then run from LDS without a memory wait between them.  The register-fragment kernel
// above has one step of prefetch and pays a memory round trip per step (15 dependent round trips per slice).
constexpr int D1F_AS = 16 * 16 + 4;              // LDS row stride of the flat slice (floats)
__host__ __device__ constexpr int d1f_lds_floats(int mt, int max_steps) { return max_steps * 8 * 256 + 16 * mt * D1F_AS; }

template <int MT>
__global__ __launch_bounds__(512) void dense1_fwd_tile_kernel(const float* __restrict__ flat, const float* __restrict__ pk,
                                                              float* __restrict__ part, int B, int ks_total, int max_steps) {
  extern __shared__ __attribute__((aligned(16))) float d1f_lds[];
  float* wls = d1f_lds;                                    // [step][8 column tiles][16 n][16 kk]
  float* als = d1f_lds + max_steps * 8 * 256;              // [16*MT rows][260]
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  const int nrow = (B + 16 * MT - 1) / (16 * MT);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int rb = j % nrow, yz = (j / nrow) * 8 + xcd;      // yz = ks * 2 + column half (blocks of one slice share an XCD)
  if (yz >= ks_total * 2) return;
  const int m0 = rb * 16 * MT, ks = yz >> 1, half = yz & 1;
  const int s0 = (ks * KSTEPS_DENSE) / ks_total, s1 = ((ks + 1) * KSTEPS_DENSE) / ks_total, steps = s1 - s0;   // <= max_steps
  // ---- stage (LDS-DMA): Wd slice, then the rows
  for (int p = wv; p < steps * 8; p += 8) {
    const int sl = p >> 3, nt = p & 7;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pk + ((size_t)(s0 + sl) * HID + half * 128 + nt * 16) * 16 + 4 * lane),
                                     (__attribute__((address_space(3))) void*)(wls + p * 256), 16, 0, 0);
  }
  for (int row = wv; row < 16 * MT; row += 8) {
    if (m0 + row < B) {                                     // wave-uniform
      if (4 * lane < 16 * steps)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(flat + (size_t)(m0 + row) * FLAT + 16 * s0 + 4 * lane),
                                         (__attribute__((address_space(3))) void*)(als + row * D1F_AS), 16, 0, 0);
    } else {
      *reinterpret_cast<f32x4*>(&als[row * D1F_AS + 4 * lane]) = zero4();
    }
  }
  __syncthreads();                                           // vmcnt(0) precedes the barrier: the DMA has landed
  f32x4 acc[MT][2];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) acc[mi][0] = acc[mi][1] = zero4();
  const float* bp = wls + (wv * 16 + r) * 16 + 4 * g;
  const float* ap = als + r * D1F_AS + 4 * g;
  for (int sl = 0; sl < steps; ++sl) {
    const f32x4 w = ld4(bp + sl * 8 * 256);
    f32x4 a[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) a[mi] = ld4(ap + mi * 16 * D1F_AS + 16 * sl);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[mi][t & 1] = mfma(a[mi][t], w[t], acc[mi][t & 1]);
  }
  float* out = part + ((size_t)ks * B) * HID + half * 128 + wv * 16 + r;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mr = m0 + mi * 16 + 4 * g + q;
      if (mr < B) out[(size_t)mr * HID] = acc[mi][0][q] + acc[mi][1][q];
    }
}

__host__ inline int dense1_fwd_blocks(int B, int ks_total, int mt) {
  const int nrow = (B + 16 * mt - 1) / (16 * mt);
  return 8 * nrow * ((ks_total * 2 + 7) / 8);
}

// ------------------------------------------------------------------ heads (+ loss)
// One wave per sample: d1 = relu(sum_ks part + bd); z = d1 Wp + bp; v = d1 Wv + bv; softmax;
// TRAIN adds the A3C loss terms, the head gradients dz, dv and the gradient that
// flows back into the hidden layer, dd1 = 1[d1>0] (dz Wp^T + dv Wv^T) -- the wave already holds Wp, Wv, d1.
struct HeadArgs {
  const float* part; int ks; int B; int A;
  const float* bd; const float* wv; const float* bv; const float* wp; const float* bp;
  float* d1; float* z; float* p; float* v;
  // train only
  const float* y_r; const float* act; float* dz; float* dv; float* lossrow; float* dd1;
  float beta, log_eps, min_policy; int log_softmax;
};

// NR rows per wave (rows[j] < 0: none; wave-uniform): every load of every row is requested before the first reduction
template <bool TRAIN, int AMAX, int NR>
__device__ __forceinline__ void heads_rows(const HeadArgs& h, const int (&rows)[NR], int lane) {
  // this lane's 4 hidden units x A policy weights are 4*A contiguous floats: issue every load up front
  float wreg[4][AMAX];
  const float* wp = h.wp + (size_t)(4 * lane) * h.A;
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int o = 0; o < AMAX; ++o) wreg[q][o] = o < h.A ? wp[q * h.A + o] : 0.f;
  const f32x4 wv4 = ld4(h.wv + 4 * lane);
  const f32x4 bd4 = ld4(h.bd + 4 * lane);
  // everything the tail needs is requested now, with the slab loads, not after the reductions (one round trip, not three)
  const bool mine = lane < h.A;
  const float bp_mine = mine ? h.bp[lane] : 0.f;
  const float bv0 = h.bv[0];
  float y[NR], a[NR];
  f32x4 dr[NR];
  const size_t kstride = (size_t)h.B * HID;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    y[j] = a[j] = 0.f;
    dr[j] = bd4;
    if (rows[j] < 0) continue;
    const int b = rows[j];
    if (TRAIN) {
      y[j] = h.y_r[b];
      a[j] = mine ? h.act[(size_t)b * h.A + lane] : 0.f;
    }
    const float* pp = h.part + (size_t)b * HID + 4 * lane;
    // dense_ks() picks at most 22 slices: all of them in flight at once, summed in slice order
    f32x4 t[22];
#pragma unroll
    for (int i = 0; i < 22; ++i) t[i] = i < h.ks ? ld4(pp + (size_t)i * kstride) : zero4();
#pragma unroll
    for (int i = 0; i < 22; ++i) dr[j] += t[i];
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (rows[j] < 0) continue;
    const int b = rows[j];
    f32x4 d = dr[j];
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = fmaxf(d[q], 0.f);
    *reinterpret_cast<f32x4*>(h.d1 + (size_t)b * HID + 4 * lane) = d;

    const float v = wave_sum(d[0] * wv4[0] + d[1] * wv4[1] + d[2] * wv4[2] + d[3] * wv4[3]) + bv0;
    float zpart[AMAX];
#pragma unroll
    for (int o = 0; o < AMAX; ++o) zpart[o] = d[0] * wreg[0][o] + d[1] * wreg[1][o] + d[2] * wreg[2][o] + d[3] * wreg[3][o];
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1)
#pragma unroll
      for (int o = 0; o < AMAX; ++o) zpart[o] += __shfl_xor(zpart[o], sh, 64);
    float zmine = -INFINITY;
#pragma unroll
    for (int o = 0; o < AMAX; ++o)
      if (lane == o && o < h.A) zmine = zpart[o] + bp_mine;
    const float zmax = wave_max(zmine);
    const float e = mine ? expf(zmine - zmax) : 0.f;
    const float esum = wave_sum(e);
    const float s = e / esum;
    const float denom = 1.0f + h.min_policy * (float)h.A;
    const float p = h.log_softmax ? s : (s + h.min_policy) / denom;
    if (mine) {
      h.z[(size_t)b * h.A + lane] = zmine;
      h.p[(size_t)b * h.A + lane] = p;
    }
    if (lane == 0) h.v[b] = v;
    if (TRAIN) {
      const float yy = y[j], aa = a[j];
      const float adv = yy - v;
      float dz, c1, c2;
      if (h.log_softmax) {
        const float ls = mine ? (zmine - zmax) - logf(esum) : 0.f;
        const float lsel = wave_sum(ls * aa);
        const float ent = wave_sum(mine ? s * ls : 0.f);
        const float asum = wave_sum(aa);
        c1 = lsel * adv;
        c2 = -h.beta * ent;
        dz = -adv * (aa - s * asum) + h.beta * s * (ls - ent);
      } else {
        const float sel = wave_sum(mine ? p * aa : 0.f);
        const float logp = logf(fmaxf(p, h.log_eps));
        c1 = logf(fmaxf(sel, h.log_eps)) * adv;
        c2 = -h.beta * wave_sum(mine ? logp * p : 0.f);
        const float gsel = sel >= h.log_eps ? 1.0f / sel : 0.f;
        const float gp = -(adv * gsel) * aa + h.beta * (logp + (p >= h.log_eps ? 1.0f : 0.f));
        const float gs = mine ? gp / denom : 0.f;
        const float dot = wave_sum(gs * s);
        dz = s * (gs - dot);
      }
      if (mine) h.dz[(size_t)b * h.A + lane] = dz;
      {
        const float dvv = v - yy;
        f32x4 dd = {dvv * wv4[0], dvv * wv4[1], dvv * wv4[2], dvv * wv4[3]};
#pragma unroll
        for (int o = 0; o < AMAX; ++o) {
          const float dzo = __shfl(dz, o, 64);          // 0 for lanes >= A
#pragma unroll
          for (int q = 0; q < 4; ++q) dd[q] += dzo * wreg[q][o];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dd[q] = d[q] > 0.f ? dd[q] : 0.f;
        *reinterpret_cast<f32x4*>(h.dd1 + (size_t)b * HID + 4 * lane) = dd;
      }
      if (lane == 0) {
        h.dv[b] = v - yy;
        h.lossrow[(size_t)b * 3 + 0] = c1;
        h.lossrow[(size_t)b * 3 + 1] = c2;
        h.lossrow[(size_t)b * 3 + 2] = 0.5f * (yy - v) * (yy - v);
      }
    }
  }
}

template <bool TRAIN, int AMAX>
__global__ __launch_bounds__(256) void heads_kernel(HeadArgs h) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // one wave per sample; 1..4 waves per workgroup
  if (b >= h.B) return;   // wave-uniform
  const int rows[1] = {b};
  heads_rows<TRAIN, AMAX, 1>(h, rows, threadIdx.x & 63);
}

// ------------------------------------------------------------------ heads backward (weight gradients)
// Runs as extra blocks of dense1_dw_kernel (no launch of its own):
// role o in [0,A]: dW[k] = sum_b d1[b][k] dhead[b] (o<A: dWp[:,o], dbp[o]; o==A: dWv, dbv), LDS fold, fixed order
// role A+1:        losses[c] = sum_b lossrow[b][c]
struct HeadBwdArgs {
  int B; int A;
  const float* d1; const float* dz; const float* dv; const float* lossrow;
  float* g_wp; float* g_bp; float* g_wv; float* g_bv; float* losses;
};

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ void heads_bwd_role(const HeadBwdArgs& h, int role) {
  __shared__ float sh[4];
  __shared__ f32x4 sacc[4][64];
  const int k = threadIdx.x;
  if (role <= h.A) {
    // thread = (4 hidden units kq, batch residue bg): float4 rows of d1, partial sums folded through LDS
    const int o = role;
    const bool isv = o == h.A;
    const int kq = k & 63, bg = k >> 6;
    f32x4 acc = zero4();
    float bsum = 0.f;
    for (int b0 = bg; b0 < h.B; b0 += 32) {
      f32x4 dd[8];
      float gh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = b0 + 4 * i;
        const bool ok = b < h.B;
        dd[i] = ok ? ld4(h.d1 + (size_t)b * HID + 4 * kq) : zero4();
        gh[i] = ok ? (isv ? h.dv[b] : h.dz[(size_t)b * h.A + o]) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += dd[i] * gh[i];
    }
    for (int b = k; b < h.B; b += 256) bsum += isv ? h.dv[b] : h.dz[(size_t)b * h.A + o];
    sacc[bg][kq] = acc;
    bsum = block_sum_256(bsum, sh);   // contains the barriers that also publish sacc
    if (bg == 0) {
      const f32x4 tot = (sacc[0][kq] + sacc[1][kq]) + (sacc[2][kq] + sacc[3][kq]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (isv) h.g_wv[4 * kq + q] = tot[q];
        else h.g_wp[(size_t)(4 * kq + q) * h.A + o] = tot[q];
      }
    }
    if (k == 0) {
      if (isv) h.g_bv[0] = bsum; else h.g_bp[o] = bsum;
    }
  } else {
    for (int c = 0; c < 3; ++c) {
      float part = 0.f;
      for (int b = k; b < h.B; b += 256) part += h.lossrow[(size_t)b * 3 + c];
      part = block_sum_256(part, sh);
      if (k == 0) h.losses[c] = part;
    }
  }
}

// loss sums alone (Network.log evaluates a batch without a backward pass: NetworkVP.py:259-265)
__global__ __launch_bounds__(256) void loss_sum_kernel(HeadBwdArgs hb) { heads_bwd_role(hb, hb.A + 1); }

// ------------------------------------------------------------------ dense1 backward: dWd = flat^T dd1
// M = 3872 (kidx), N = 256, contraction over the batch.  Wave tile 32 x 32 (2 x 2 MFMA tiles).
// grid.x = 121 row blocks, grid.y = 2, wave -> 32-column group (grid.y*4 + wave).
// Row block 0 also produces dbd[n] = sum_b dd1[b][n].  Blocks with blockIdx.x >= 121 (grid.y == 0 only) carry the
// head weight gradients and the loss sums (heads_bwd_role).
__device__ __forceinline__ void dense1_dw_body(const float* __restrict__ flat, const float* __restrict__ dd1,
                                                        float* __restrict__ g_wd, float* __restrict__ g_bd, int B,
                                                        HeadBwdArgs hb, int bx, int by, int gx) {
  (void)gx;
  if (bx >= FLAT / 32) {   // block-uniform
    if (by == 0) heads_bwd_role(hb, bx - FLAT / 32);
    return;
  }
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int m0 = bx * 32;
  const int n0 = (by * 4 + (threadIdx.x >> 6)) * 32;
  f32x4 acc[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};
  float bs0 = 0.f, bs1 = 0.f;
  const int nsteps = (B + 15) >> 4;
  constexpr int G = 4;   // 16-row batch steps per operand group (64 dword loads in flight behind 64 MFMAs)
  float ca[G][4][2], cb[G][4][2];
  auto fetch = [&](int s, float (&a)[4][2], float (&bb)[4][2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int b = 16 * s + 4 * g + t;
      const bool ok = b < B;                 // also false for steps past the end
      const float* fr = flat + (size_t)(ok ? b : 0) * FLAT + m0 + r;
      const float* dr = dd1 + (size_t)(ok ? b : 0) * HID + n0 + r;
      a[t][0] = ok ? fr[0] : 0.f;
      a[t][1] = ok ? fr[16] : 0.f;
      bb[t][0] = ok ? dr[0] : 0.f;
      bb[t][1] = ok ? dr[16] : 0.f;
    }
  };
#pragma unroll
  for (int i = 0; i < G; ++i) fetch(i, ca[i], cb[i]);
  for (int s = 0; s < nsteps; s += G) {
    float na[G][4][2], nb[G][4][2];
#pragma unroll
    for (int i = 0; i < G; ++i) fetch(s + G + i, na[i], nb[i]);
#pragma unroll
    for (int i = 0; i < G; ++i) {
      if (s + i < nsteps) {                  // wave-uniform
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bs0 += cb[i][t][0];
          bs1 += cb[i][t][1];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma(ca[i][t][mi], cb[i][t][ni], acc[mi][ni]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ca[i][t][0] = na[i][t][0]; ca[i][t][1] = na[i][t][1];
        cb[i][t][0] = nb[i][t][0]; cb[i][t][1] = nb[i][t][1];
      }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        g_wd[(size_t)(m0 + mi * 16 + 4 * g + q) * HID + n0 + ni * 16 + r] = acc[mi][ni][q];
  if (bx == 0) {
    bs0 += __shfl_xor(bs0, 16, 64); bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 16, 64); bs1 += __shfl_xor(bs1, 32, 64);
    if (g == 0) {
      g_bd[n0 + r] = bs0;
      g_bd[n0 + 16 + r] = bs1;
    }
  }
}

// ------------------------------------------------------------------ dense1 backward: dn2 = (dd1 Wd^T) * 1[n2>0]
// M = B, N = 3872, K = 256.  Both operands are k-contiguous (16-byte loads).  Wave tile (16*MT) x 32: the two Wd
// row fragments a wave loads are reused for MT row tiles (MT = 1 at small batch for parallelism, 4 at large batch
// where re-reading Wd once per 16 rows was the bottleneck).  bx = column block of 32, by*4 + wave = row block.
template <int MT>
__device__ __forceinline__ void dense1_dx_body(const float* __restrict__ dd1, const float* __restrict__ wd,
                                               const float* __restrict__ n2, float* __restrict__ dn2, int B, int bx, int by,
                                               int gx) {
  (void)gx;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int n0 = bx * 32;
  const int m0 = (by * 4 + (threadIdx.x >> 6)) * 16 * MT;
  if (m0 >= B) return;   // wave-uniform
  const float* b0 = wd + (size_t)(n0 + r) * HID + 4 * g;
  const float* b1 = wd + (size_t)(n0 + 16 + r) * HID + 4 * g;
  // K = 256 only: the Wd fragments (32 x 16-byte loads) are issued up front and stay in registers
  f32x4 w0[16], w1[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    w0[s] = ld4(b0 + 16 * s);
    w1[s] = ld4(b1 + 16 * s);
  }
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int mbase = m0 + 16 * mi;
    if (mbase >= B) break;   // wave-uniform
    const int row = mbase + r;
    const bool valid = row < B;
    const float* arow = dd1 + (size_t)(valid ? row : 0) * HID + 4 * g;
    f32x4 a[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = valid ? ld4(arow + 16 * s) : zero4();
    float mask0[4], mask1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mr = mbase + 4 * g + q;
      const size_t o = (size_t)(mr < B ? mr : 0) * FLAT + n0 + r;
      mask0[q] = n2[o];
      mask1[q] = n2[o + 16];
    }
    f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = mfma(a[s][t], w0[s][t], acc0);
        acc1 = mfma(a[s][t], w1[s][t], acc1);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mr = mbase + 4 * g + q;
      if (mr < B) {
        const size_t o = (size_t)mr * FLAT + n0 + r;
        dn2[o] = mask0[q] > 0.f ? acc0[q] : 0.f;
        dn2[o + 16] = mask1[q] > 0.f ? acc1[q] : 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------ conv2 backward: dW2 partials
// dW2[kidx][o] = sum_q patch(q)[kidx] dn2[q][o]; kidx = (u*4+v)*16 + c, q over B*121 pixels.
// m-tile mt = patch position (u,v), lane row = channel c.  Workgroup = (sample group, quarter of the 16 patch
// positions): a sample's n1 (zero-padded 24x24x16) and dn2 (121 px padded to 128) are staged in LDS, each wave
// owns one patch position x 32 columns and contracts over the 128 pixel slots (q = 16 s + 4 t + g, so that the
// four lane groups read neighbouring pixels).  The next sample's loads are in flight during the MFMAs.
// part[group][8192 + 32]: the +32 tail is the group's db2 (quarter 0, wave 0).  Reduced by slab_reduce_kernel.
constexpr int SLAB2 = 256 * 32 + 32;
constexpr int C2DW_IMG = C2_PW * C2_PW * C1;     // 9216 floats
constexpr int C2DW_DN = 128 * C2;                // 4096 floats

// PF: the next sample's loads are in flight during the MFMAs (52 VGPRs held across them); without, a sample's loads are
// requested when its turn comes -- the form for grids that give every workgroup ONE sample (gx = B)
template <bool PF = true>
__device__ __forceinline__ void conv2_dw_body(const float* __restrict__ n1, const float* __restrict__ dn2,
                                                          float* __restrict__ part, int B, int bx, int by, int gx,
                                                          float* __restrict__ lds) {   // lds: C2DW_IMG + C2DW_DN floats
  float* img = lds;
  float* dnl = lds + C2DW_IMG;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  const int mt = by * 4 + wv, u = mt >> 2, v = mt & 3;
  f32x4 acc0 = zero4(), acc1 = zero4();
  float bs0 = 0.f, bs1 = 0.f;
  f32x4 simg[9], sdn[4];
  auto fetch = [&](int b) {
    const float* nb = n1 + (size_t)b * N1S;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = threadIdx.x + 256 * i;           // float4 index: pixel = idx>>2, channel quad = idx&3
      const int px = idx >> 2, row = px / C2_PW, col = px - row * C2_PW;
      const int yy = row - 1, xx = col - 1;
      const bool ok = (unsigned)yy < (unsigned)O1 && (unsigned)xx < (unsigned)O1;
      simg[i] = ok ? ld4(nb + (yy * O1 + xx) * C1 + (idx & 3) * 4) : zero4();
    }
    const float* db = dn2 + (size_t)b * FLAT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + 256 * i;           // float4 index into [128 px][32]
      sdn[i] = idx < FLAT / 4 ? ld4(db + 4 * idx) : zero4();
    }
  };
  int b = bx;
  if (PF && b < B) fetch(b);
  for (; b < B; b += gx) {
    if (!PF) fetch(b);
    __syncthreads();                                   // everyone is done reading the previous sample
#pragma unroll
    for (int i = 0; i < 9; ++i) *reinterpret_cast<f32x4*>(&img[(threadIdx.x + 256 * i) * 4]) = simg[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&dnl[(threadIdx.x + 256 * i) * 4]) = sdn[i];
    __syncthreads();
    if (PF && b + gx < B) fetch(b + gx);
    // operands of step s+1 are fetched and pinned in registers of their own before the MFMAs of step s
    float oa[2][4], ob[2][4][2];
    auto load_step = [&](int s, float (&a)[4], float (&bb)[4][2]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int q = 16 * s + 4 * t + g;
        const int qc = q < P2 ? q : 0;                 // slots 121..127 carry dn2 = 0
        const int i = qc / O2, j = qc - i * O2;
        a[t] = img[((2 * i + u) * C2_PW + 2 * j + v) * C1 + r];
        bb[t][0] = dnl[q * C2 + r];
        bb[t][1] = dnl[q * C2 + 16 + r];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) { pin(a[t]); pin(bb[t][0]); pin(bb[t][1]); }
    };
    load_step(0, oa[0], ob[0]);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 1 < 8) load_step(s + 1, oa[(s + 1) & 1], ob[(s + 1) & 1]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bs0 += ob[s & 1][t][0];
        bs1 += ob[s & 1][t][1];
        acc0 = mfma(oa[s & 1][t], ob[s & 1][t][0], acc0);
        acc1 = mfma(oa[s & 1][t], ob[s & 1][t][1], acc1);
      }
    }
  }
  float* out = part + (size_t)bx * SLAB2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    out[(mt * 16 + 4 * g + q) * C2 + r] = acc0[q];
    out[(mt * 16 + 4 * g + q) * C2 + 16 + r] = acc1[q];
  }
  if (mt == 0) {
    bs0 += __shfl_xor(bs0, 16, 64); bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 16, 64); bs1 += __shfl_xor(bs1, 32, 64);
    if (g == 0) {
      out[256 * 32 + r] = bs0;
      out[256 * 32 + 16 + r] = bs1;
    }
  }
}

// ------------------------------------------------------------------ conv2 backward: dn1 (transposed conv as a gather)
// dn1[b,y,x,c] = 1[n1>0] sum_{ua,va in {0,1}} sum_o dn2[b,i,j,o] W2[u,v,c,o],
//   i = ((y+1)>>1) - ua, u = ((y+1)&1) + 2 ua (same for x): K = 4 taps * 32 = 128, N = 16.
// Pixels are grouped by parity class (py,px) so that a 16-pixel tile shares one weight sub-matrix.
// Workgroup = (sample, class): the sample's dn2 goes to LDS once with a zero row/column for i = -1 / j = -1,
// the class's 128 x 16 weight sub-matrix goes to LDS in fragment order, and the 7-8 tiles of the class are
// dealt to the 4 waves.  No atomics, no col buffer; every dn1 element is written exactly once.
constexpr int C2DX_CELL = C2 + 4;            // floats per pixel cell of the dn2 image in LDS: 36, not 32 -- with 128-byte cells the
                                            // 16 pixels of a tile hit 2 bank groups with their 16-byte patch reads (8-way conflict)
constexpr int C2DX_DN = 12 * 12 * C2DX_CELL; // padded dn2 image: (i+1, j+1), 5184 floats
constexpr int C2DX_W = 32 * 64;             // weight fragments of one class

// the 7-8 tiles of parity class (PY, PX) of one sample, dealt to the 4 waves of a wave group (wg = wave index in the group)
template <int PY, int PX>
__device__ __forceinline__ void conv2_dx_tiles(const float* __restrict__ dnl, const float* __restrict__ wl,
                                               const float* __restrict__ n1b, float* __restrict__ d1b, int wg) {
  constexpr int NY = PY ? 10 : 11, NX = PX ? 10 : 11, CNT = NY * NX;
  constexpr int NTILE = (CNT + 15) / 16;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  for (int tile = wg; tile < NTILE; tile += 4) {
    const int mc = tile * 16 + r;
    const int mm = mc < CNT ? mc : 0;
    const int ya = mm / NX, xa = mm - ya * NX;
    const int ih = (2 * ya + PY + 1) >> 1, jh = (2 * xa + PX + 1) >> 1;
    // ReLU mask of the 4 pixels this lane will write
    float mask[4];
    int off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mr = tile * 16 + 4 * g + q;
      const int m2 = mr < CNT ? mr : 0;
      const int ya2 = m2 / NX, xa2 = m2 - ya2 * NX;
      off[q] = ((2 * ya2 + PY) * O1 + (2 * xa2 + PX)) * C1 + r;
      mask[q] = n1b[off[q]];
    }
    f32x4 a[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
      a[s] = ld4(dnl + ((ih - (s >> 2) + 1) * 12 + (jh - ((s >> 1) & 1) + 1)) * C2DX_CELL + (s & 1) * 16 + 4 * g);
    f32x4 acc0 = zero4(), acc1 = zero4();
    f32x4 wq[2][2];                                      // the weight fragments of an s pair (one 16-byte read per step), one pair ahead
    auto load_w = [&](int s, f32x4 (&w2)[2]) {
      w2[0] = ld4(wl + (s * 64 + lane) * 4);
      w2[1] = ld4(wl + ((s + 1) * 64 + lane) * 4);
      pin(w2[0]); pin(w2[1]);
    };
    load_w(0, wq[0]);
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
      if (s + 2 < 8) load_w(s + 2, wq[((s >> 1) + 1) & 1]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = mfma(a[s][t], wq[(s >> 1) & 1][0][t], acc0);
        acc1 = mfma(a[s + 1][t], wq[(s >> 1) & 1][1][t], acc1);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tile * 16 + 4 * g + q < CNT) d1b[off[q]] = mask[q] > 0.f ? acc0[q] + acc1[q] : 0.f;
  }
}

// Workgroup (8 waves) = (sample, row parity PY): the sample's zero-bordered dn2 image is staged ONCE for the two column
// parities, each with its own weight sub-matrix and its own group of 4 waves (one class per workgroup staged the image
// four times per sample and left the kernel at 14 % MFMA utilisation).
template <int PY>
__device__ __forceinline__ void conv2_dx_pair(const float* __restrict__ dn2, const float* __restrict__ w,
                                              const float* __restrict__ n1, float* __restrict__ dn1, int b,
                                              float* __restrict__ dnl, float* __restrict__ wl) {
  const float* db = dn2 + (size_t)b * FLAT;
  f32x4 sd[3];
  // w: the packed conv12/w (pack_w2dx_kernel order): classes (PY,0), (PY,1) are 2 x 2048 contiguous floats
  const f32x4 sw0 = ld4(w + (PY * 2 + 0) * C2DX_W + 4 * threadIdx.x), sw1 = ld4(w + (PY * 2 + 1) * C2DX_W + 4 * threadIdx.x);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = threadIdx.x + 512 * i;                 // float4 index into [12][12][8]
    const int pi = idx >> 3, pr = pi / 12, pc = pi - pr * 12;
    const bool ok = idx < 12 * 12 * 8 && pr >= 1 && pc >= 1;
    sd[i] = ok ? ld4(db + ((pr - 1) * O2 + (pc - 1)) * C2 + (idx & 7) * 4) : zero4();
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = threadIdx.x + 512 * i;
    if (idx < 12 * 12 * 8) *reinterpret_cast<f32x4*>(&dnl[(idx >> 3) * C2DX_CELL + (idx & 7) * 4]) = sd[i];
  }
  *reinterpret_cast<f32x4*>(&wl[4 * threadIdx.x]) = sw0;
  *reinterpret_cast<f32x4*>(&wl[C2DX_W + 4 * threadIdx.x]) = sw1;
  __syncthreads();
  const float* n1b = n1 + (size_t)b * N1S;
  float* d1b = dn1 + (size_t)b * N1S;
  const int wv = threadIdx.x >> 6;
  if (wv < 4) conv2_dx_tiles<PY, 0>(dnl, wl, n1b, d1b, wv);
  else conv2_dx_tiles<PY, 1>(dnl, wl + C2DX_W, n1b, d1b, wv - 4);
}

__device__ __forceinline__ void conv2_dx_body(const float* __restrict__ dn2, const float* __restrict__ w,
                                              const float* __restrict__ n1, float* __restrict__ dn1, int B, int bx, int by,
                                              int gx) {
  __shared__ __attribute__((aligned(16))) float c2dx_lds[C2DX_DN + 2 * C2DX_W];
  float* dnl = c2dx_lds;
  float* wl = c2dx_lds + C2DX_DN;
  (void)gx;
  if (bx >= B) return;   // block-uniform guard: bx = sample
  if (by == 0) conv2_dx_pair<0>(dn2, w, n1, dn1, bx, dnl, wl);   // block-uniform: by = row parity
  else conv2_dx_pair<1>(dn2, w, n1, dn1, bx, dnl, wl);
}

// ------------------------------------------------------------------ dense1 backward, both gradients, LDS-tiled
// Workgroup (16 waves) = 16 consecutive flat columns k0..k0+15 (242 workgroups = one round on the 256 CUs):
//   dWd[k0+i][n]  = sum_b flat[b][k0+i] dd1[b][n]            M = 16, N = 256, contraction over the batch
//   dn2[b][k0+i]  = 1[n2[b][k0+i] > 0] sum_n dd1[b][n] Wd[k0+i][n]      M = B, N = 16, contraction over 256
// Both need ALL of dd1, which is what the two separate bodies above re-read from L2 once per wave (31 MB of fragment
// loads at batch 128): here dd1 (a chunk of up to 128 rows, 128 KB), the 16 Wd rows (16 KB) and the 16 flat columns
// (8 KB) go to LDS once per workgroup -- dd1 and Wd by LDS-DMA: a row is 1 KB = one wave instruction, so the rows can be
// padded to 260 floats, which keeps the row-strided reads of the batch contraction (rows 4 apart for the lane groups of
// a 32-lane LDS phase: 4 * 260 = 16 mod 32 banks) off each other; the flat columns through registers, transposed to
// [column][row], so that a lane's four batch rows of a step are one 16-byte read -- and every MFMA operand is an LDS read.  Waves 0-7 own two 16-column tiles of dWd each (accumulated over the chunks), waves 8-15
// one 16-row tile of dn2 each: 64 MFMAs per wave and chunk, four waves per SIMD.  Workgroup 0 also sums dd1 into dbd.
// Blocks past the 242 tiles carry the head weight gradients and the loss sums.
constexpr int D1B_ROWS = 128;                    // batch rows per chunk
constexpr int D1B_DS = HID + 4;                  // padded row stride of dd1 / Wd in LDS (floats)
constexpr int D1B_COLS = 16;                     // flat columns per workgroup
constexpr int D1B_NS = D1B_ROWS + 4;             // row stride of the transposed flat columns [16][128] in LDS
constexpr int D1B_LDS_FLOATS = D1B_ROWS * D1B_DS + D1B_COLS * D1B_DS + D1B_COLS * D1B_NS;   // 39552 floats = 158,208 B
constexpr int D1B_TILES = FLAT / D1B_COLS;       // 242
// Tail area: the 5,632 bytes a CU's LDS has left beside a chunk hold the dd1 rows and flat columns of up to 5 rows PAST the first
// chunk (a "batch = 128" trainer assembles 129 .. 133 rows), requested together with the first chunk's staging loads; the
// second chunk is then staged LDS -> LDS instead of waiting for a second memory round trip behind the first chunk's MFMAs.
constexpr int D1B_TAIL_ROWS = 5;
constexpr int D1B_TAIL_FLOATS = D1B_TAIL_ROWS * D1B_DS + D1B_TAIL_ROWS * D1B_COLS;          // 1380 floats
constexpr int D1B_LDS_FLOATS_TAIL = D1B_LDS_FLOATS + D1B_TAIL_FLOATS;                       // 40932 floats = 163,728 B
static_assert(D1B_LDS_FLOATS_TAIL * sizeof(float) <= 160 * 1024, "dense1_bwd_tile: the tail area must fit the CU's 160 KB");

// head weight gradients / loss sums for a 1024-thread block (same arithmetic order as heads_bwd_role up to the fold width)
template <bool UPD>
__device__ __forceinline__ void heads_bwd_role_wide(const HeadBwdArgs& h, int role, float* lds, const FusedUpd& u) {
  f32x4* sacc = reinterpret_cast<f32x4*>(lds);            // [16][64]
  float* sh = lds + 16 * 64 * 4;                           // [16]
  const int k = threadIdx.x, kq = k & 63, bg = k >> 6;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((k & 63) == 0) sh[bg] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i];
    return t;
  };
  if (role <= h.A) {
    const int o = role;
    const bool isv = o == h.A;
    f32x4 acc = zero4();
    float bsum = 0.f;
    for (int b0 = bg; b0 < h.B; b0 += 128) {
      f32x4 dd[8];
      float gh[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = b0 + 16 * i;
        const bool ok = b < h.B;
        dd[i] = ok ? ld4(h.d1 + (size_t)b * HID + 4 * kq) : zero4();
        gh[i] = ok ? (isv ? h.dv[b] : h.dz[(size_t)b * h.A + o]) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += dd[i] * gh[i];
    }
    for (int b = k; b < h.B; b += 1024) bsum += isv ? h.dv[b] : h.dz[(size_t)b * h.A + o];
    __syncthreads();                                        // a previous role's readers are done with sacc
    sacc[bg * 64 + kq] = acc;
    bsum = block_sum(bsum);                                 // its barriers also publish sacc
    if (bg == 0) {
      f32x4 tot = zero4();
#pragma unroll
      for (int i = 0; i < 16; ++i) tot += sacc[i * 64 + kq];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (isv) h.g_wv[4 * kq + q] = tot[q];
        else h.g_wp[(size_t)(4 * kq + q) * h.A + o] = tot[q];
        if (UPD) fused_rmsprop(u, isv ? OFF_WV + 4 * kq + q : OFF_WP + (int64_t)(4 * kq + q) * h.A + o, tot[q]);
      }
    }
    if (k == 0) {
      if (isv) h.g_bv[0] = bsum; else h.g_bp[o] = bsum;
      if (UPD) fused_rmsprop(u, isv ? OFF_BV : off_bp(h.A) + o, bsum);
    }
  } else {
    for (int c = 0; c < 3; ++c) {
      float part = 0.f;
      for (int b = k; b < h.B; b += 1024) part += h.lossrow[(size_t)b * 3 + c];
      part = block_sum(part);
      if (k == 0) h.losses[c] = part;
    }
  }
}

struct Dense1TileArgs {
  const float* n2; const float* dd1; const float* wd; float* g_wd; float* g_bd; float* dn2; int B;
  HeadBwdArgs hb; int role_blocks;
  int tail_lds;      // host side only: the TAIL form is launched
  FusedUpd upd;      // on: dense1/w, dense1/b and the head parameters are stepped here (see FusedUpd)
};

// UPD: the launch also steps the parameters whose gradients it completes (FusedUpd; a.upd.on says the same at run time):
// 1 all of them, 2 all but dense1/w (FusedUpd::defer_wd: the next launch, conv_bwd, steps it).  A template argument rather
// than a run-time switch so that the forms are separate kernels to a profiler: their HBM traffic differs by the optimizer's
// 16 MB (profiles/: tools/pmc_table.py lists them as separate rows).
// TAIL: D1B_ROWS < B <= D1B_ROWS + D1B_TAIL_ROWS and the launch carries D1B_LDS_FLOATS_TAIL floats of LDS: the rows past the
// chunk are worked on out of the tail area BESIDE the chunk -- one more step of the batch contraction for the dWd waves, four
// MFMAs of the rows' dn2 tile for every wave, folded in the epilogue behind barriers that are there anyway -- instead of as a
// second chunk (two more barriers, a second staging pass and 64 dependent MFMAs on one wave: 9.0 -> 10.9 us at 132 rows)
template <int UPD, bool TAIL = false>
__global__ __launch_bounds__(1024) void dense1_bwd_tile_kernel(Dense1TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float d1b_lds[];
  if ((int)blockIdx.x >= D1B_TILES) {                       // block-uniform: the head roles, dealt round-robin
    for (int role = blockIdx.x - D1B_TILES; role < a.hb.A + 2; role += a.role_blocks) heads_bwd_role_wide<(UPD != 0)>(a.hb, role, d1b_lds, a.upd);
    return;
  }
  float* dds = d1b_lds;                                     // [128][260]  dd1 rows of the chunk
  float* wds = dds + D1B_ROWS * D1B_DS;                     // [16][260]   Wd rows k0..k0+15
  float* n2s = wds + D1B_COLS * D1B_DS;                     // [16][132]   flat columns k0..k0+15 of the chunk's rows, transposed
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  // column tile of this workgroup: tiles 2j and 2j + 1 -- the two halves of every 128-byte line of the flat rows -- are
  // eight workgroups apart, i.e. on one XCD (blocks are dealt round-robin over the 8 XCDs), so each line of n2 comes out of
  // HBM once instead of once per half (a placement hint only; the last two of the 242 tiles keep their own index)
  const int bid = blockIdx.x;
  const int tile = bid < 240 ? 16 * (bid >> 4) + 2 * (bid & 7) + ((bid >> 3) & 1) : bid;
  const int k0 = tile * D1B_COLS;
  const int B = a.B;
  f32x4 accw[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};   // dWd tiles n-tile 2 wv, 2 wv + 1 (waves 0-7), two chains each
  float bs0 = 0.f, bs1 = 0.f;
  // fused update: the `ms` slot and the weights of the 16 x 256 tile this workgroup will step in its epilogue are requested
  // NOW, with the staging loads, and arrive under the MFMAs -- fetched in the epilogue they were a memory round trip and a
  // half in front of 4 MB of stores (13.4 us with the update against 8.6 us bare)
  const int64_t uidx = (int64_t)(k0 + (threadIdx.x >> 6)) * HID + (threadIdx.x & 63) * 4;
  constexpr bool upd_wd = UPD == 1;
  f32x4 pre_ms = zero4(), pre_th = zero4();
  if (upd_wd) {
    pre_ms = ld4(a.upd.ms + OFF_WD + uidx);
    pre_th = ld4(a.upd.tin + OFF_WD + uidx);
  }
  float* tdd = d1b_lds + D1B_LDS_FLOATS;                         // [5][260]  dd1 rows 128.. (tail area, a.tail_lds only)
  float* tn2 = tdd + D1B_TAIL_ROWS * D1B_DS;                     // [5][16]   their flat columns k0..k0+15
  constexpr bool tail = TAIL;
  const int trows = tail ? B - D1B_ROWS : 0;                     // 1 .. 5 rows worked on out of the tail area, beside the chunk
  f32x4 tacc = zero4();                                          // this wave's partial of their dn2 tile (tail only)
  for (int c0 = 0; c0 < B - trows; c0 += D1B_ROWS) {
    const int rows = B - trows - c0 < D1B_ROWS ? B - trows - c0 : D1B_ROWS;      // real rows of this chunk
    const int prow = (rows + 15) & ~15;                          // padded to whole MFMA tiles
    if (c0) __syncthreads();                                     // everyone is done reading the previous chunk
    {
    // ---- stage: one 1 KB wave instruction per dd1 / Wd row, one per 16 rows of the flat columns
    for (int row = wv; row < prow; row += 16) {
      if (row < rows)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.dd1 + (size_t)(c0 + row) * HID + 4 * lane),
                                         (__attribute__((address_space(3))) void*)(dds + row * D1B_DS), 16, 0, 0);
      else
        *reinterpret_cast<f32x4*>(&dds[row * D1B_DS + 4 * lane]) = zero4();
    }
    if (c0 == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wd + (size_t)(k0 + wv) * HID + 4 * lane),
                                       (__attribute__((address_space(3))) void*)(wds + wv * D1B_DS), 16, 0, 0);
    if (c0 == 0 && tail) {                                       // rows 128.. : requested now, used by the second chunk
      if (wv < B - D1B_ROWS)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.dd1 + (size_t)(D1B_ROWS + wv) * HID + 4 * lane),
                                         (__attribute__((address_space(3))) void*)(tdd + wv * D1B_DS), 16, 0, 0);
    }
    if (threadIdx.x < 4 * D1B_ROWS) {                            // thread -> (row tid/4, float4 tid%4), stored transposed
      const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
      const f32x4 v = row < rows ? ld4(a.n2 + (size_t)(c0 + row) * FLAT + k0 + 4 * c) : zero4();
#pragma unroll
      for (int j = 0; j < 4; ++j) n2s[(4 * c + j) * D1B_NS + row] = v[j];
    } else if (c0 == 0 && tail && (int)threadIdx.x - 4 * D1B_ROWS < 4 * (B - D1B_ROWS)) {
      const int i = threadIdx.x - 4 * D1B_ROWS, row = i >> 2, c = i & 3;   // waves 8..: the tail rows' flat columns, row-major
      *reinterpret_cast<f32x4*>(&tn2[row * D1B_COLS + 4 * c]) = ld4(a.n2 + (size_t)(D1B_ROWS + row) * FLAT + k0 + 4 * c);
    }
    }
    __syncthreads();                                             // vmcnt(0) precedes the barrier: the DMA has landed
    if (wv < 8) {
      // ---- dWd: contraction over the chunk's rows, b = 16 s + 4 t + g
      const int n0 = wv * 32;
      // operands of step s+1 are fetched (and pinned in registers of their own) before the MFMAs of step s
      f32x4 av[2];
      float bw[2][4][2];
      auto load_step = [&](int s, f32x4& a, float (&b)[4][2]) {
        a = ld4(n2s + r * D1B_NS + 16 * s + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int bb = 16 * s + 4 * g + t;
          b[t][0] = dds[bb * D1B_DS + n0 + r];
          b[t][1] = dds[bb * D1B_DS + n0 + 16 + r];
        }
        pin(a);
#pragma unroll
        for (int t = 0; t < 4; ++t) { pin(b[t][0]); pin(b[t][1]); }
      };
      const int nstep = prow / 16;
      load_step(0, av[0], bw[0]);
      for (int s = 0; s < nstep; s += 2) {
        // two steps per trip so that the double buffer is indexed by constants
        if (s + 1 < nstep) load_step(s + 1, av[1], bw[1]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bs0 += bw[0][t][0];
          bs1 += bw[0][t][1];
          accw[0][t & 1] = mfma(av[0][t], bw[0][t][0], accw[0][t & 1]);
          accw[1][t & 1] = mfma(av[0][t], bw[0][t][1], accw[1][t & 1]);
        }
        if (s + 1 < nstep) {
          if (s + 2 < nstep) load_step(s + 2, av[0], bw[0]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            bs0 += bw[1][t][0];
            bs1 += bw[1][t][1];
            accw[0][t & 1] = mfma(av[1][t], bw[1][t][0], accw[0][t & 1]);
            accw[1][t & 1] = mfma(av[1][t], bw[1][t][1], accw[1][t & 1]);
          }
        }
      }
      if (trows) {
        // the tail rows' step of the batch contraction, b = 128 + 4 g + t, out of the tail area (rows past B are zeros): the
        // same MFMAs in the same order as a second chunk's only step
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int bb = 4 * g + t;
          const bool ok = bb < trows;
          const float av1 = ok ? tn2[bb * D1B_COLS + r] : 0.f;
          const float b0 = ok ? tdd[bb * D1B_DS + n0 + r] : 0.f, b1 = ok ? tdd[bb * D1B_DS + n0 + 16 + r] : 0.f;
          bs0 += b0;
          bs1 += b1;
          accw[0][t & 1] = mfma(av1, b0, accw[0][t & 1]);
          accw[1][t & 1] = mfma(av1, b1, accw[1][t & 1]);
        }
      }
    }
    if (trows) {                                                 // block-uniform
      // ---- dn2 of the tail rows: one tile, its contraction cut over the 16 waves (k = 16 wv + 4 g + t), folded in the epilogue
      const f32x4 xa = r < trows ? ld4(tdd + r * D1B_DS + 4 * g + 16 * wv) : zero4(), xw = ld4(wds + r * D1B_DS + 4 * g + 16 * wv);
#pragma unroll
      for (int t = 0; t < 4; ++t) tacc = mfma(xa[t], xw[t], tacc);
    }
    if (c0 && prow == 16) {                                      // block-uniform
      // ---- dn2 of a chunk of <= 16 rows (the rows past the first 128): its one tile is 64 dependent MFMAs -- 0.85 us on ONE
      // wave while fifteen wait -- so the contraction is cut over the 16 waves, four MFMAs each (k = 16 wv + 4 g + t), and
      // the 16 partial tiles are folded in wave order through the LDS rows 16.. of dds that such a chunk leaves unused
      float* ps = dds + 16 * D1B_DS;                             // [16 waves][16 rows][16 columns]
      {
        const f32x4 xa = ld4(dds + r * D1B_DS + 4 * g + 16 * wv), xw = ld4(wds + r * D1B_DS + 4 * g + 16 * wv);
        f32x4 acc = zero4();
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma(xa[t], xw[t], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) ps[wv * 256 + (4 * g + q) * 16 + r] = acc[q];
      }
      __syncthreads();
      if (threadIdx.x < 256) {
        const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += ps[w * 256 + row * 16 + col];
        if (row < rows) a.dn2[(size_t)(c0 + row) * FLAT + k0 + col] = n2s[col * D1B_NS + row] > 0.f ? v : 0.f;
      }
    } else if (wv >= 8) {
      // ---- dn2: one 16-row tile per wave, contraction over the 256 hidden units, k = 16 s + 4 g + t
      const int m0 = (wv - 8) * 16;
      if (m0 < prow) {                                           // wave-uniform
        const float* ap = dds + (m0 + r) * D1B_DS + 4 * g;
        const float* bp = wds + r * D1B_DS + 4 * g;
        f32x4 acc0 = zero4(), acc1 = zero4();
        f32x4 xa[2][2], xw[2][2];
        auto load_pair = [&](int s, f32x4 (&a)[2], f32x4 (&w)[2]) {
          a[0] = ld4(ap + 16 * s); w[0] = ld4(bp + 16 * s); a[1] = ld4(ap + 16 * s + 16); w[1] = ld4(bp + 16 * s + 16);
          pin(a[0]); pin(w[0]); pin(a[1]); pin(w[1]);
        };
        load_pair(0, xa[0], xw[0]);
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
          if (s + 2 < 16) load_pair(s + 2, xa[((s >> 1) + 1) & 1], xw[((s >> 1) + 1) & 1]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc0 = mfma(xa[(s >> 1) & 1][0][t], xw[(s >> 1) & 1][0][t], acc0);
            acc1 = mfma(xa[(s >> 1) & 1][1][t], xw[(s >> 1) & 1][1][t], acc1);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + 4 * g + q;
          if (row < rows)
            a.dn2[(size_t)(c0 + row) * FLAT + k0 + r] = n2s[r * D1B_NS + row] > 0.f ? acc0[q] + acc1[q] : 0.f;
        }
      }
    }
  }
  // ---- epilogue: the 16 x 256 dWd tile goes through LDS so that gradient, ms and weights move as whole 1 KB rows
  // (16 bytes per lane) -- the accumulator layout would touch them in 64-byte pieces -- and all 16 waves take part
  __syncthreads();                                             // every wave is done with the staged operands
  float* gt = d1b_lds;                                         // [16][256]
  float* tps = dds + 16 * D1B_DS;                              // [16 waves][16 rows][16 columns], past gt
  if (trows) {
#pragma unroll
    for (int q = 0; q < 4; ++q) tps[wv * 256 + (4 * g + q) * 16 + r] = tacc[q];
  }
  if (wv < 8) {
    const int n0 = wv * 32;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) gt[(4 * g + q) * HID + n0 + ni * 16 + r] = accw[ni][0][q] + accw[ni][1][q];
    if (tile == 0) {
      bs0 += __shfl_xor(bs0, 16, 64); bs0 += __shfl_xor(bs0, 32, 64);
      bs1 += __shfl_xor(bs1, 16, 64); bs1 += __shfl_xor(bs1, 32, 64);
      if (g == 0) {
        a.g_bd[n0 + r] = bs0;
        a.g_bd[n0 + 16 + r] = bs1;
        if (UPD) {
          fused_rmsprop(a.upd, OFF_BD + n0 + r, bs0);
          fused_rmsprop(a.upd, OFF_BD + n0 + 16 + r, bs1);
        }
      }
    }
  }
  __syncthreads();
  if (trows && threadIdx.x >= 768) {                           // waves 12..15 fold the tail rows' dn2 tile, in wave order
    const int i = threadIdx.x - 768, row = i >> 4, col = i & 15;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += tps[w * 256 + row * 16 + col];
    if (row < trows) a.dn2[(size_t)(D1B_ROWS + row) * FLAT + k0 + col] = tn2[row * D1B_COLS + col] > 0.f ? v : 0.f;
  }
  {
    const int row = threadIdx.x >> 6, c4 = (threadIdx.x & 63) * 4;
    const int64_t idx = uidx;
    const f32x4 gv = ld4(gt + row * HID + c4);
    *reinterpret_cast<f32x4*>(a.g_wd + idx) = gv;
    if (upd_wd) {
      const FusedUpd& u = a.upd;
      f32x4 m = pre_ms;
      const f32x4 th = pre_th;
      f32x4 mo = u.mu != 0.f ? ld4(u.mom + OFF_WD + idx) : zero4();
      f32x4 tn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        m[e] += (gv[e] * gv[e] - m[e]) * u.omr;
        float step = (gv[e] * u.lr) / sqrtf(u.eps + m[e]);
        if (u.mu != 0.f) { step = mo[e] * u.mu + step; mo[e] = step; }
        tn[e] = th[e] - step;
      }
      *reinterpret_cast<f32x4*>(u.ms + OFF_WD + idx) = m;
      if (u.mu != 0.f) *reinterpret_cast<f32x4*>(u.mom + OFF_WD + idx) = mo;
      *reinterpret_cast<f32x4*>(u.tout + OFF_WD + idx) = tn;
      *reinterpret_cast<f32x4*>(gt + row * HID + c4) = tn;       // for the fragment-ordered copy below
    }
  }
  if (upd_wd) {
    __syncthreads();
    // the fragment-ordered copy dense1_fwd reads: rows 4 kq .. 4 kq + 3 of column n are 16 contiguous bytes there
    const int n = threadIdx.x & 255, kq = threadIdx.x >> 8;
    const f32x4 v = {gt[(4 * kq) * HID + n], gt[(4 * kq + 1) * HID + n], gt[(4 * kq + 2) * HID + n], gt[(4 * kq + 3) * HID + n]};
    *reinterpret_cast<f32x4*>(a.upd.pk + ((size_t)tile * HID + n) * 16 + 4 * kq) = v;
  }
}

// ---- launchable forms.  The two gradients of a layer (weights / inputs) only share inputs, so they are also
// offered as ONE launch whose block range is split between the two bodies: they then run side by side and
// fill the chip, which neither does alone at small batch (cross-stream events cost more than they save here).
__global__ __launch_bounds__(256) void dense1_dw_kernel(const float* __restrict__ flat, const float* __restrict__ dd1,
                                                        float* __restrict__ g_wd, float* __restrict__ g_bd, int B,
                                                        HeadBwdArgs hb) {
  dense1_dw_body(flat, dd1, g_wd, g_bd, B, hb, blockIdx.x, blockIdx.y, gridDim.x);
}
__global__ __launch_bounds__(256) void dense1_dx_kernel(const float* __restrict__ dd1, const float* __restrict__ wd,
                                                        const float* __restrict__ n2, float* __restrict__ dn2, int B) {
  dense1_dx_body<1>(dd1, wd, n2, dn2, B, blockIdx.x, blockIdx.y, gridDim.x);
}
struct Dense1BwdArgs {
  const float* n2; const float* dd1; const float* wd; float* g_wd; float* g_bd; float* dn2; int B;
  HeadBwdArgs hb; int dw_gx; int dw_blocks; int dx_gx; int dx_mt;
};
__global__ __launch_bounds__(256) void dense1_bwd_kernel(Dense1BwdArgs a) {
  const int id = blockIdx.x;
  if (id < a.dw_blocks) dense1_dw_body(a.n2, a.dd1, a.g_wd, a.g_bd, a.B, a.hb, id % a.dw_gx, id / a.dw_gx, a.dw_gx);
  else {
    const int j = id - a.dw_blocks;
    if (a.dx_mt == 4) dense1_dx_body<4>(a.dd1, a.wd, a.n2, a.dn2, a.B, j % a.dx_gx, j / a.dx_gx, a.dx_gx);
    else if (a.dx_mt == 2) dense1_dx_body<2>(a.dd1, a.wd, a.n2, a.dn2, a.B, j % a.dx_gx, j / a.dx_gx, a.dx_gx);
    else dense1_dx_body<1>(a.dd1, a.wd, a.n2, a.dn2, a.B, j % a.dx_gx, j / a.dx_gx, a.dx_gx);
  }
}
// OCC = workgroups per CU the register budget is cut for.  2: 205 VGPRs, the next sample's loads in flight during the MFMAs:
// the form for batches whose workgroups walk several samples (B > 256).  3: 149 VGPRs, no loads held across the MFMAs
// (conv2_dw_body<false>): three workgroups' LDS (3 x 53,248 B) fit a CU, so the 4 B workgroups of a batch are one round up to 192
// rows instead of 128 -- 132 rows: 9.0 -> 7.5 us -- and it is no slower anywhere up to 256 rows (one sample per workgroup).
// Same arithmetic, same bits.
template <int OCC>
__global__ __launch_bounds__(256, OCC) void conv2_dw_kernel(const float* __restrict__ n1, const float* __restrict__ dn2,
                                                            float* __restrict__ part, int B) {
  __shared__ __attribute__((aligned(16))) float lds[C2DW_IMG + C2DW_DN];
  conv2_dw_body<OCC == 2>(n1, dn2, part, B, blockIdx.x, blockIdx.y, gridDim.x, lds);
}
__global__ __launch_bounds__(512) void conv2_dx_kernel(const float* __restrict__ dn2, const float* __restrict__ w,
                                                       const float* __restrict__ n1, float* __restrict__ dn1, int B) {
  conv2_dx_body(dn2, w, n1, dn1, B, blockIdx.x, blockIdx.y, gridDim.x);
}

// ------------------------------------------------------------------ conv1 backward: dW1 partials
// dW1[kidx][o] = sum_q patch(q)[kidx] dn1[q][o]; kidx = (u*8+v)*4+c; m-tile mt = u*2 + (v>>2),
// lane row r = (v&3)*4 + c = 16 contiguous floats of one input row.  Workgroup walks units = (sample, band of
// 3 output rows) exactly as conv1_fwd cuts them: the band's 16 padded input rows and its 63 dn1 pixels (padded
// to 64) are staged in LDS, each wave owns 4 m-tiles x 16 columns and contracts over the 64 pixel slots
// (q = 16 s + 4 t + g).  The next unit's loads are in flight during the MFMAs.
// part[workgroup][4096 + 16] (+16 = db1 partial, wave 0).  Reduced by slab_reduce_kernel.
constexpr int SLAB1 = 256 * 16 + 16;
constexpr int C1DW_IMG = C1_RIN * C1_PW * 4;     // 5632 floats
constexpr int C1DW_DN = 64 * C1;                 // 1024 floats

// bid / nblk: this workgroup's index among the nblk that share the units (= partial slabs); lds: C1DW_IMG + C1DW_DN floats
template <bool U8>
__device__ __forceinline__ void conv1_dw_body(const void* __restrict__ x, const float* __restrict__ dn1,
                                              float* __restrict__ part, int nunits, int bid, int nblk, float* __restrict__ lds) {
  float* img = lds;
  float* dnl = lds + C1DW_IMG;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, mg = threadIdx.x >> 6;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  float bs = 0.f;
  f32x4 simg[U8 ? 1 : 6], sdn;
  // uint8 states: a band's 16 rows are 16 x 21 pieces of 16 bytes (four pixels) -- 336 wide loads for 256 threads instead of
  // 1,408 dword loads (six per thread), converted when they are written to LDS; the four padding columns of the image, which
  // no piece covers, are zeroed once.  Same values in LDS either way.
  constexpr int U8_PIECES = C1_RIN * (IMG / 4);              // 336
  uint4 raw[2];
  auto fetch = [&](int unit) {
    const int b = unit / 7, band = unit - b * 7;
    const int y_base = 4 * C1_HB * band - 2;
    if constexpr (U8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = threadIdx.x + 256 * i;
        const int row = t / (IMG / 4), grp = t - row * (IMG / 4);
        const int yy = y_base + row;
        const bool ok = t < U8_PIECES && (unsigned)yy < (unsigned)IMG;
        raw[i] = ok ? *reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(x) + (size_t)b * XS + (size_t)(yy * IMG + 4 * grp) * 4)
                    : make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);   // 128 / 128 - 1 = 0: rows outside the image
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int row = idx / C1_PW, col = idx - row * C1_PW;
        const int yy = y_base + row, xx = col - 2;
        const bool ok = idx < C1_RIN * C1_PW && (unsigned)yy < (unsigned)IMG && (unsigned)xx < (unsigned)IMG;
        simg[i] = ok ? load_px<U8>(x, b, yy * IMG + xx) : zero4();
      }
    }
    const float* db = dn1 + ((size_t)b * P1 + band * C1_HB * O1) * C1;
    sdn = threadIdx.x < C1_HB * O1 * C1 / 4 ? ld4(db + 4 * threadIdx.x) : zero4();   // 252 float4, slot 63 zero
  };
  if (U8 && threadIdx.x < 4 * C1_RIN) {                      // padding columns 0, 1, 86, 87 of the 16 rows
    const int row = threadIdx.x >> 2, c = threadIdx.x & 3;
    *reinterpret_cast<f32x4*>(&img[(row * C1_PW + (c < 2 ? c : IMG + c)) * 4]) = zero4();
  }
  int unit = bid;
  if (unit < nunits) fetch(unit);
  for (; unit < nunits; unit += nblk) {
    __syncthreads();
    if constexpr (U8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = threadIdx.x + 256 * i;
        if (t < U8_PIECES) {
          const int row = t / (IMG / 4), grp = t - row * (IMG / 4);
          float* dst = img + (row * C1_PW + 2 + 4 * grp) * 4;
          *reinterpret_cast<f32x4*>(dst) = px_from_u8(raw[i].x);
          *reinterpret_cast<f32x4*>(dst + 4) = px_from_u8(raw[i].y);
          *reinterpret_cast<f32x4*>(dst + 8) = px_from_u8(raw[i].z);
          *reinterpret_cast<f32x4*>(dst + 12) = px_from_u8(raw[i].w);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < C1_RIN * C1_PW) *reinterpret_cast<f32x4*>(&img[idx * 4]) = simg[i];
      }
    }
    *reinterpret_cast<f32x4*>(&dnl[threadIdx.x * 4]) = sdn;
    __syncthreads();
    if (unit + nblk < nunits) fetch(unit + nblk);
    // The 20 LDS operands of a step (4 pixel slots x (1 dn1 value + 4 patch values)) are read into registers of their
    // own one step AHEAD of the 16 MFMAs that consume them: left to itself hipcc re-used one register pair for every
    // read and put an lgkmcnt(0) wait in front of every second MFMA (LDS latency exposed 32 times per unit).
    float av[2][4][4], bv[2][4];
    auto load_step = [&](int s, float (&a)[4][4], float (&b)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int q = 16 * s + 4 * t + g;
        const int qc = q < C1_HB * O1 ? q : 0;          // slot 63 carries dn1 = 0
        const int il = qc / O1, j = qc - il * O1;
        b[t] = dnl[q * C1 + r];
        const float* ap = img + (4 * il) * (C1_PW * 4) + 16 * j + r;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int mt = mg * 4 + mi;                   // u = mt>>1, vh = mt&1
          a[t][mi] = ap[(mt >> 1) * (C1_PW * 4) + (mt & 1) * 16];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        pin(b[t]);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) pin(a[t][mi]);
      }
    };
    load_step(0, av[0], bv[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s + 1 < 4) load_step(s + 1, av[(s + 1) & 1], bv[(s + 1) & 1]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bs += bv[s & 1][t];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[mi] = mfma(av[s & 1][t][mi], bv[s & 1][t], acc[mi]);
      }
    }
  }
  float* out = part + (size_t)bid * SLAB1;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) out[((mg * 4 + mi) * 16 + 4 * g + q) * C1 + r] = acc[mi][q];
  if (mg == 0) {
    bs += __shfl_xor(bs, 16, 64); bs += __shfl_xor(bs, 32, 64);
    if (g == 0) out[256 * 16 + r] = bs;
  }
}
template <bool U8>
__global__ __launch_bounds__(256, 2) void conv1_dw_kernel(const void* __restrict__ x, const float* __restrict__ dn1,
                                                          float* __restrict__ part, int nunits) {
  __shared__ __attribute__((aligned(16))) float lds[C1DW_IMG + C1DW_DN];
  conv1_dw_body<U8>(x, dn1, part, nunits, blockIdx.x, gridDim.x, lds);
}
// The two weight gradients of the split path in ONE launch (batches up to 256 rows: one sample per conv2_dw workgroup).
// They only share dn1 / dn2 as inputs (conv2_dx runs before them), both are 256-thread workgroups far from any roof of a CU
// (19-33 % MFMA, < 2 TB/s) that spend most of their time waiting for their own staging loads: side by side on a CU -- three
// workgroups of either kind, 53 KB of LDS each (conv1_dw uses half of it) -- they fill each other's waits, and a kernel
// boundary goes.  The conv1_dw workgroups (two units each, the longer ones) come first in the grid.  Slabs as before.
template <bool U8>
__global__ __launch_bounds__(256, 3) void conv_dw_pair_kernel(const void* __restrict__ x, const float* __restrict__ dn1,
                                                              float* __restrict__ part1, int nunits, int nblk1,
                                                              const float* __restrict__ n1, const float* __restrict__ dn2,
                                                              float* __restrict__ part2, int B, int nch2) {
  __shared__ __attribute__((aligned(16))) float lds[C2DW_IMG + C2DW_DN];
  static_assert(C1DW_IMG + C1DW_DN <= C2DW_IMG + C2DW_DN, "conv_dw_pair: conv1_dw's image must fit conv2_dw's LDS");
  if ((int)blockIdx.x < nblk1) {                              // block-uniform
    conv1_dw_body<U8>(x, dn1, part1, nunits, blockIdx.x, nblk1, lds);
  } else {
    const int id = blockIdx.x - nblk1;
    conv2_dw_body<false>(n1, dn2, part2, B, id % nch2, id / nch2, nch2, lds);
  }
}

// ------------------------------------------------------------------ conv backward, all three gradients of a sample half
// conv2_dw, conv2_dx and conv1_dw each work sample by sample and hand dn2 -> dn1 through HBM; as three launches they cost
// three grids of load round trips and two kernel boundaries (6.1 + 5.8 + 8.6 us at batch 128).  Here a workgroup (16 waves)
// = (sample, half):
//   phase 1  the dn1 pixels it OWNS = the transposed conv of dn2, by output-pixel parity class (conv2_dx_tiles'
//            arithmetic), ReLU mask from the n1 image in LDS; the result stays in LDS (and goes to HBM only for
//            ga3c_net_fetch: compute_grads keeps it, train steps pass dn1 = nullptr).  Ownership is by pixel -- rows 0..9 and row 10 up to column 17 | the rest -- so that every
//            class of either half fits 4 tiles of 16 pixels: 16 tiles, one per wave (rows 0..10 | 11..20 gave one half 17
//            tiles, i.e. one wave a second tile of two pixels that everybody waited for).  The ragged class lists live in
//            an LDS table written once per workgroup.
//   phase 2  dW2 partial over its conv2 pixels (60 | 61, conv_stack_fwd's cut): wave = (pair of patch positions, half of
//            the 64 pixel slots): every (dn2, n1) operand fetch and its pixel -> (row, column) arithmetic feeds 4 MFMAs
//   phase 3  dW1 partial over n1 rows 0..10 | 10..20 in bands of 3 rows (a pixel of row 10 the half does not own is a
//            zero in its dn1 image): wave = (4 m-tiles, quarter of the band's 64 pixel slots), conv1_dw_kernel's blocking;
//            the half's 48 padded x rows (66 KB) are requested by LDS-DMA behind the first barrier, in flight under
//            phases 1 and 2, so that the bands run back to back without a barrier
//   (a first version gave every wave one m-tile / one position: one operand pair and one index computation per MFMA made
//   phases 2 and 3 VALU-bound, 26.8 us against 22.9 us for the three launches)
// The K-split partial sums of the waves are folded through LDS once, at the end.  Partial dW2 / dW1 (+ bias
// tails) go to slab2 / slab1 in conv2_dw's / conv1_dw's layout; slab_reduce_kernel is unchanged.
constexpr int CB_N1IMG = CS_N1ROWS * C2_PW * C1;             // 5376 floats
constexpr int CB_DN2IMG = C2DX_DN;                           // 5184: cells of 36 floats (bank spread, see C2DX_CELL)
constexpr int CB_W2 = 4 * C2DX_W;                            // 8192
constexpr int CB_DN1 = 11 * O1 * C1;                         // 3696: n1 rows 0..10 | 10..20, [row][col][16]
constexpr int CB_XROWS = 4 * 11 + 4;                         // x rows 4 r0 - 2 .. 4 r0 + 45 of the half's 11 n1 rows
constexpr int CB_XIMG = CB_XROWS * C1_PW * 4;                // 16896 floats: the whole x image of the half, padded
constexpr int CB_LDS_FLOATS = CB_N1IMG + CB_DN2IMG + CB_W2 + CB_DN1 + CB_XIMG + 256;   // 39600 floats = 158,400 B
constexpr int CB_ROW10_CUT = 18;                             // row 10: columns < 18 belong to half 0

// slot m of parity class (py, px) of half h -> (y << 8) | x, or -1
__device__ __forceinline__ int cb_class_pixel(int h, int py, int px, int m) {
  const int nx = px ? 10 : 11;
  if (h == 0) {
    if (m < 5 * nx) return ((py + 2 * (m / nx)) << 8) | (2 * (m % nx) + px);
    const int e = m - 5 * nx;
    return (py == 0 && e < 9) ? ((10 << 8) | (2 * e + px)) : -1;          // row 10: x = px, px + 2, ... <= 17
  }
  const int np = py == 0 ? (px ? 1 : 2) : 0;                               // row 10: x = 18, 20 | 19
  if (m < np) return (10 << 8) | (CB_ROW10_CUT + px + 2 * m);
  const int e = m - np, ya = e / nx;
  return ya < 5 ? (((py ? 11 : 12) + 2 * ya) << 8) | (2 * (e % nx) + px) : -1;
}

// The optimizer step of dense1/w riding in conv_bwd (FusedUpd::defer_wd).  In dense1_bwd_tile's epilogue the step was 20 MB
// of loads and stores that every workgroup issued at the same moment, behind the MFMA phase: 4.7 us of a 14.3-us kernel
// with nothing to hide them under.  conv_bwd is bound by MFMA and LDS and leaves the vector-memory path idle after its
// staging: each workgroup steps one 16-row group of dense1/w (242 groups, 16 KB each) -- operands requested after phase 1
// (in flight during phase 2), arithmetic and stores in front of the barrier
// before phase 3, where early waves wait for late ones anyway; the stores drain under phase 3, which reads only LDS.  Thread ->
// (column n = tid % 256, row quad q = tid / 256): rows 16 s + 4 q .. + 3 of column n, which are ONE 16-byte piece of the
// fragment-ordered copy (pack_wd_kernel); the row-major arrays move as 256-byte runs per wave instruction.  Same arithmetic,
// element by element, as fused_rmsprop / rmsprop_one.
// Stores that go THROUGH the XCD's L2 to memory as they are issued (device scope: sc1) instead of staying dirty in it until
// the end-of-kernel write-back: a kernel's end waits for its L2s to drain, so bytes stored early and written through cost
// the kernel nothing, bytes left dirty cost it their write-back whenever they were stored.
#ifdef GA3C_STORE_THROUGH
__device__ __forceinline__ void store_through(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_through4(float* p, f32x4 v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) __hip_atomic_store(p + j, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
__device__ __forceinline__ void store_through(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_through4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
#endif
struct WdStep { float g[4], m[4], t[4]; };
// q0: workgroups of fewer than 1024 threads walk the group's four row quads in passes (q = q0 + tid / 256)
__device__ __forceinline__ void wd_step_load(const FusedUpd& u, const float* __restrict__ g_wd, int s, WdStep& w, int q0 = 0) {
  const int n = threadIdx.x & 255, q = q0 + (threadIdx.x >> 8);
  const int64_t i0 = (int64_t)(16 * s + 4 * q) * HID + n;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w.g[j] = g_wd[i0 + j * HID];
    w.m[j] = u.ms[OFF_WD + i0 + j * HID];
    w.t[j] = u.tin[OFF_WD + i0 + j * HID];
  }
}
__device__ __forceinline__ void wd_step_apply(const FusedUpd& u, int s, const WdStep& w, int q0 = 0) {
  const int n = threadIdx.x & 255, q = q0 + (threadIdx.x >> 8);
  const int64_t i0 = OFF_WD + (int64_t)(16 * s + 4 * q) * HID + n;
  f32x4 tn;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float m = w.m[j];
    m += (w.g[j] * w.g[j] - m) * u.omr;
    store_through(u.ms + i0 + j * HID, m);
    float step = (w.g[j] * u.lr) / sqrtf(u.eps + m);
    if (u.mu != 0.f) {
      step = u.mom[i0 + j * HID] * u.mu + step;
      store_through(u.mom + i0 + j * HID, step);
    }
    tn[j] = w.t[j] - step;
    store_through(u.tout + i0 + j * HID, tn[j]);
  }
  store_through4(u.pk + ((size_t)s * HID + n) * 16 + 4 * q, tn);
}

// The same step riding in conv2_dx, for batches beyond the fused conv_bwd's 128 rows (the split path: conv2_dw, conv2_dx,
// conv1_dw).  There dense1_bwd_tile<1> carried it in its epilogue: 14.8 us in a 132-row step against 10.4 without.  conv2_dx
// is 264 .. 512 workgroups of which a CU can hold three, bound by its MFMA chains, with 12 MB of traffic in 8.5 us: the 242
// row groups go to 121 x 2 workgroups of their own at the FRONT of the grid (dispatched first: their loads and stores are
// under way while the gradient workgroups stage and compute), two passes of two row quads each.
constexpr int C2DX_WD_BLOCKS = KSTEPS_DENSE / 2;
static_assert(KSTEPS_DENSE % 2 == 0, "conv2_dx_wd_kernel deals the row groups of dense1/w to pairs of workgroups");
__global__ __launch_bounds__(512) void conv2_dx_wd_kernel(const float* __restrict__ dn2, const float* __restrict__ w,
                                                          const float* __restrict__ n1, float* __restrict__ dn1, int B,
                                                          const float* __restrict__ g_wd, FusedUpd u, int wd_first) {
  const int bxw = wd_first ? (int)blockIdx.x : (int)blockIdx.x - B;        // index among the step's block columns
  const int bxc = wd_first ? (int)blockIdx.x - C2DX_WD_BLOCKS : (int)blockIdx.x;
  if (bxw >= 0 && bxw < C2DX_WD_BLOCKS) {                      // block-uniform
    const int s = 2 * bxw + blockIdx.y;
    WdStep w0, w1;
    wd_step_load(u, g_wd, s, w0, 0);
    wd_step_load(u, g_wd, s, w1, 2);
    wd_step_apply(u, s, w0, 0);
    wd_step_apply(u, s, w1, 2);
    return;
  }
  conv2_dx_body(dn2, w, n1, dn1, B, bxc, blockIdx.y, gridDim.x - C2DX_WD_BLOCKS);
}

// WD: the launch also applies the optimizer step u to dense1/w, whose gradient g_wd is complete since the previous launch
// (FusedUpd::defer_wd; a template argument so that a profiler lists the two forms apart: their traffic differs by 24 MB);
// otherwise neither g_wd nor u is touched
template <bool U8, bool WD = false>
__global__ __launch_bounds__(1024) void conv_bwd_kernel(const void* __restrict__ x, const float* __restrict__ n1,
                                                       const float* __restrict__ dn2, const float* __restrict__ w2pk,
                                                       float* __restrict__ dn1, float* __restrict__ slab2,
                                                       float* __restrict__ slab1, int B, const float* __restrict__ g_wd,
                                                       FusedUpd u) {
  extern __shared__ __attribute__((aligned(16))) float cb_lds[];
  GA3C_STAMP(0);
  float* n1img = cb_lds;                                     // [14][24][16]  n1 rows n1org .. n1org+13, cols -1..22
  float* dnimg = n1img + CB_N1IMG;                           // [12][12][32]  dn2 (i+1, j+1), zero row / column 0
  float* w2l = dnimg + CB_DN2IMG;                            // [4 classes][32][64]
  float* dn1l = w2l + CB_W2;                                 // [11 rows][21][16]
  float* xb = dn1l + CB_DN1;                                 // [48][88][4]
  int* ptab = reinterpret_cast<int*>(xb + CB_XIMG);          // [4 classes][64 slots]
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
  if ((int)blockIdx.x >= 2 * B) return;                      // block-uniform
  constexpr bool step_wd = WD;
  const bool step_mine = step_wd && (int)blockIdx.x < KSTEPS_DENSE;
  WdStep wst;
  int grp, h;
  cs_sample_half(blockIdx.x, B, grp, h);                     // both halves of a sample on one XCD (see cs_sample_half)
  const int q0 = h ? CS_C2CUT : 0, c2npix = h ? P2 - CS_C2CUT : CS_C2CUT;   // conv2 pixels of this half
  const int c2r0 = h ? 5 : 0, n1org = 2 * c2r0 - 1;          // n1 image row 0 holds n1 row n1org
  const int r0 = h ? 10 : 0;                                 // first n1 row of its dn1 image; 11 rows, 4 bands
  constexpr int NROWS = 11, NBAND = 4;
  // ---- once per workgroup: the packed conv12/w, zeros in the cells no DMA lane will ever write, the class table
  for (int p = wv; p < CB_W2 / 256; p += 16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w2pk + p * 256 + 4 * lane),
                                     (__attribute__((address_space(3))) void*)(w2l + p * 256), 16, 0, 0);
  for (int i = threadIdx.x; i < CB_N1IMG / 4; i += 1024) {
    const int cell = i >> 2, row = cell / C2_PW, col = cell - row * C2_PW;
    if (!((unsigned)(n1org + row) < (unsigned)O1 && (unsigned)(col - 1) < (unsigned)O1)) *reinterpret_cast<f32x4*>(&n1img[4 * i]) = zero4();
  }
  for (int i = threadIdx.x; i < 12 * 12 * 8; i += 1024) {    // only the border cells: the others are written sample by sample
    const int cell = i >> 3, pr = cell / 12, pc = cell - pr * 12;
    if (pr < 1 || pc < 1) *reinterpret_cast<f32x4*>(&dnimg[cell * C2DX_CELL + 4 * (i & 7)]) = zero4();
  }
  if (threadIdx.x < 256) ptab[threadIdx.x] = cb_class_pixel(h, threadIdx.x >> 7, (threadIdx.x >> 6) & 1, threadIdx.x & 63);
  f32x4 acc2[4] = {zero4(), zero4(), zero4(), zero4()};                 // dW2: tile t, row r = row 64 u + 4 r + t
  f32x4 acc1[4] = {zero4(), zero4(), zero4(), zero4()};                 // dW1: tile t, row r = row 64 mg + 4 r + t
  float bs2 = 0.f, bs1 = 0.f;
  auto stage_x = [&](int b) {                                // the half's x rows of sample b -> xb (phase 3 reads them)
    const int y_base = 4 * r0 - 2;
    for (int idx = threadIdx.x; idx < CB_XROWS * C1_PW; idx += 1024) {
      const int row = idx / C1_PW, col = idx - row * C1_PW;
      const int yy = y_base + row, xx = col - 2;
      const bool ok = (unsigned)yy < (unsigned)IMG && (unsigned)xx < (unsigned)IMG;
      if (U8) {
        *reinterpret_cast<f32x4*>(&xb[idx * 4]) = ok ? load_px<true>(x, b, yy * IMG + xx) : zero4();
      } else if (ok) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const float*>(x) + (size_t)b * XS + (size_t)(yy * IMG + xx) * 4),
                                         (__attribute__((address_space(3))) void*)(xb + (size_t)(idx - lane) * 4), 16, 0, 0);
      } else {
        *reinterpret_cast<f32x4*>(&xb[idx * 4]) = zero4();
      }
    }
  };
  {
    const int b = grp;                                       // one sample per workgroup: the grid is 2 * B (B <= 128, one round)
    // ---- stage what phases 1 and 2 read: n1 image (cells of 64 B, 4 lanes each), dn2 image (cells of 128 B, 8 lanes each);
    // clear the dn1 image (row 10's pixels of the other half stay zero)
    const float* n1b = n1 + (size_t)b * N1S;
    const float* db = dn2 + (size_t)b * FLAT;
    for (int i = threadIdx.x; i < CB_N1IMG / 4; i += 1024) {
      const int cell = i >> 2, row = cell / C2_PW, col = cell - row * C2_PW;
      const int yy = n1org + row, xx = col - 1;
      if ((unsigned)yy < (unsigned)O1 && (unsigned)xx < (unsigned)O1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(n1b + (yy * O1 + xx) * C1 + 4 * (i & 3)),
                                         (__attribute__((address_space(3))) void*)(n1img + (size_t)(i - lane) * 4), 16, 0, 0);
    }
    if (threadIdx.x < FLAT / 4) {                            // 968 float4 through registers into the padded cells
      const int q = threadIdx.x >> 3, i2 = q / O2, j2 = q - i2 * O2;
      *reinterpret_cast<f32x4*>(&dnimg[((i2 + 1) * 12 + j2 + 1) * C2DX_CELL + 4 * (threadIdx.x & 7)]) = ld4(db + 4 * threadIdx.x);
    }
    if (threadIdx.x < CB_DN1 / 4) *reinterpret_cast<f32x4*>(&dn1l[4 * threadIdx.x]) = zero4();
    if (U8) stage_x(b);                                      // 17 KB of uint8 per half: with the other loads, converted on the way
    __syncthreads();                                         // vmcnt(0) precedes the barrier: the images have landed
    GA3C_STAMP(1);
    if (!U8) stage_x(b);                                     // 66 KB by LDS-DMA, in flight during phases 1 and 2
    GA3C_STAMP(2);
    // ---- phase 1: the 16 tiles of its dn1 pixels; wave = (class wv / 4, tile wv % 4)
    {
      const int cls = wv >> 2, tile = wv & 3;
      const float* wl = w2l + cls * C2DX_W;
      const int* tab = ptab + cls * 64 + tile * 16;
      const int e0 = tab[r];
      const int e = e0 >= 0 ? e0 : ptab[cls * 64];           // pad slots compute the class's first pixel and store nothing
      const int y = e >> 8, xq = e & 255;
      const int ih = (y + 1) >> 1, jh = (xq + 1) >> 1;
      f32x4 a[8];
#pragma unroll
      for (int s = 0; s < 8; ++s)
        a[s] = ld4(dnimg + ((ih - (s >> 2) + 1) * 12 + (jh - ((s >> 1) & 1) + 1)) * C2DX_CELL + (s & 1) * 16 + 4 * g);
      f32x4 c0 = zero4(), c1 = zero4();
#pragma unroll
      for (int s = 0; s < 8; s += 2) {
        f32x4 w0 = ld4(wl + (s * 64 + lane) * 4), w1 = ld4(wl + ((s + 1) * 64 + lane) * 4);   // a step's four B fragments: one read
        pin(w0); pin(w1);                                    // (4 waves per SIMD: the other waves cover this fetch)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          c0 = mfma(a[s][t], w0[t], c0);
          c1 = mfma(a[s + 1][t], w1[t], c1);
        }
      }
      const int4 e4 = *reinterpret_cast<const int4*>(&tab[4 * g]);
      const int em[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (em[q] >= 0) {
          const int y2 = em[q] >> 8, x2 = em[q] & 255;
          const float keep = n1img[((y2 - n1org) * C2_PW + x2 + 1) * C1 + r];
          const float val = keep > 0.f ? c0[q] + c1[q] : 0.f;
          dn1l[((y2 - r0) * O1 + x2) * C1 + r] = val;
          if (dn1) dn1[(size_t)b * N1S + (y2 * O1 + x2) * C1 + r] = val;    // nullptr: nobody will fetch it (train steps)
        }
      }
    }
    GA3C_STAMP(3);
    // Requested HERE: phase 1 ends in a vmcnt(0) (its LDS reads may alias the x DMA as far as the compiler knows), so
    // anything requested earlier is waited for there -- with the staging loads the step delayed the first barrier, with
    // the x rows the end of phase 1, by the 2.4 us its 12 MB take (conv_bwd 17.4 -> 20.3 us both times).
    if (step_mine) wd_step_load(u, g_wd, blockIdx.x, wst);
    // ---- phase 2: dW2 partial over conv2 pixels q0 .. q0+c2npix-1 (slots to 64 carry dn2 = 0)
    // wave = (patch row u, column half nh, K half kh: pixel slots 32 kh .. 32 kh + 31).  The 64 rows of dW2 that belong to
    // patch row u -- positions (u, 0..3) x 16 channels -- are 64 CONTIGUOUS floats of the n1 image at every conv2 pixel, so
    // a lane fetches ONE 16-byte piece (floats 4r .. 4r+3) and uses it as the A operand of four MFMAs: tile t, row r holds
    // dW2 row 64 u + 4 r + t.  One wide LDS read per four MFMAs instead of four narrow ones; same sums in the same order.
    {
      const int u = wv & 3, nh = (wv >> 2) & 1, kh = wv >> 3;
      const float* abase = n1img + (u * C2_PW) * C1 + 4 * r;
#pragma unroll 1
      for (int s0 = 0; s0 < 8; s0 += 4) {
        f32x4 a[4];
        float bq[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int q = 32 * kh + 4 * (s0 + s) + g;
          const bool ok = q < c2npix;
          const int qq = q0 + (ok ? q : 0);
          const int i2 = qq / O2, j2 = qq - i2 * O2;
          a[s] = ld4(abase + ((2 * (i2 - c2r0)) * C2_PW + 2 * j2) * C1);
          bq[s] = dnimg[(ok ? ((i2 + 1) * 12 + j2 + 1) * C2DX_CELL : 0) + nh * 16 + r];       // cell (0,0) is zero
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) { pin(a[s]); pin(bq[s]); }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          bs2 += bq[s];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc2[t] = mfma(a[s][t], bq[s], acc2[t]);
        }
      }
    }
    GA3C_STAMP(4);
    if (step_mine) wd_step_apply(u, blockIdx.x, wst);        // early waves' arithmetic under the late waves' MFMAs; the stores drain under phase 3
    __syncthreads();                                         // dn1 of the half is complete in LDS; its x rows have landed
    GA3C_STAMP(5);
    // ---- phase 3: dW1 partial, band by band (no barrier between bands: the whole x image is resident);
    // wave = (row group mg: dW1 rows 64 mg .. 64 mg + 63 = patch rows 2 mg, 2 mg + 1, K quarter kq: pixel slots 16 kq ..
    // 16 kq + 15 of the band).  As in phase 2 the 64 rows are two runs of 32 contiguous floats of the x image at every
    // pixel (8 columns x 4 channels of one patch row), a lane fetches floats 4r .. 4r+3 of them as ONE 16-byte piece and
    // feeds four MFMAs with it: tile t, row r = dW1 row 64 mg + 4 r + t.
    {
      const int mg = wv & 3, kq = wv >> 2;
      const int aoff = (2 * mg + (r >> 3)) * (C1_PW * 4) + ((4 * r) & 31);
      f32x4 av[2][4];
      float bv[2][4];
      auto load_band = [&](int k, f32x4 (&a4)[4], float (&b4)[4]) {
        const int brow0 = C1_HB * k;                         // first image row of the band
        const float* img = xb + (4 * brow0) * (C1_PW * 4) + aoff;
        const int bpix = (NROWS - brow0 < C1_HB ? NROWS - brow0 : C1_HB) * O1;       // 63; 42 in the last band
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int q = 16 * kq + 4 * t + g;
          const bool ok = q < bpix;
          const int qc = ok ? q : 0;
          const int il = qc / O1, j = qc - il * O1;
          b4[t] = ok ? dn1l[(brow0 * O1 + q) * C1 + r] : 0.f;
          a4[t] = ld4(img + (4 * il) * (C1_PW * 4) + 16 * j);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { pin(b4[t]); pin(a4[t]); }
      };
      auto band_mfma = [&](const f32x4 (&a4)[4], const float (&b4)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          bs1 += b4[t];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc1[mi] = mfma(a4[t][mi], b4[t], acc1[mi]);
        }
      };
      static_assert(NBAND % 2 == 0, "bands are processed in pairs");
      load_band(0, av[0], bv[0]);
#pragma unroll 1
      for (int k = 0; k < NBAND; k += 2) {                   // the next band's operands are fetched under this band's MFMAs
        load_band(k + 1, av[1], bv[1]);
        band_mfma(av[0], bv[0]);
        if (k + 2 < NBAND) load_band(k + 2, av[0], bv[0]);
        band_mfma(av[1], bv[1]);
      }
    }
    GA3C_STAMP(6);
  }
  // ---- fold the K-split partial sums of the waves through LDS (fixed order), then write this workgroup's slab pair:
  // waves 0-7 the dW2 tiles, waves 8-11 the dW1 tiles, wave 12 the bias sums
  {
    f32x4* red2 = reinterpret_cast<f32x4*>(cb_lds);          // everything staged above is dead now
    f32x4* red1 = red2 + 16 * 256;
    float* redb = cb_lds + 2 * 16 * 1024;                    // bs2 of waves 0, 4, 8, 12 (u = 0: [kh][nh]); bs1 of waves 0, 4, 8, 12 (mg = 0: [kq])
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      red2[wv * 256 + 64 * t + lane] = acc2[t];
      red1[wv * 256 + 64 * t + lane] = acc1[t];
    }
    if ((wv & 3) == 0) {
      redb[(wv >> 2) * 64 + lane] = bs2;                     // wave 4 nh + 8 kh -> slot nh + 2 kh
      redb[256 + (wv >> 2) * 64 + lane] = bs1;
    }
    __syncthreads();
    float* o2 = slab2 + (size_t)(2 * grp + h) * SLAB2;      // slab order = (sample, half), whatever workgroup computed it:
    float* o1 = slab1 + (size_t)(2 * grp + h) * SLAB1;      // the reduction's order, hence its bits, does not depend on the placement
    if (wv < 8) {                                            // (u, nh) = (wv & 3, wv >> 2): K half 0 (wave wv) + K half 1 (wave wv + 8)
      const int u = wv & 3, nh = wv >> 2;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 s0 = red2[wv * 256 + 64 * t + lane] + red2[(wv + 8) * 256 + 64 * t + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q)                          // tile row 4 g + q = dW2 row 64 u + 4 (4 g + q) + t
          o2[(64 * u + 4 * (4 * g + q) + t) * C2 + nh * 16 + r] = s0[q];
      }
    } else if (wv < 12) {                                    // rows 64 mg ..: K quarters = waves mg, mg + 4, mg + 8, mg + 12
      const int mg = wv - 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 tot = red1[mg * 256 + 64 * t + lane];
#pragma unroll
        for (int kq = 1; kq < 4; ++kq) tot += red1[(mg + 4 * kq) * 256 + 64 * t + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) o1[(64 * mg + 4 * (4 * g + q) + t) * C1 + r] = tot[q];
      }
    } else if (wv == 12) {
      float ta = redb[lane] + redb[128 + lane], tb = redb[64 + lane] + redb[192 + lane];
      float t1 = (redb[256 + lane] + redb[320 + lane]) + (redb[384 + lane] + redb[448 + lane]);
      ta += __shfl_xor(ta, 16, 64); ta += __shfl_xor(ta, 32, 64);
      tb += __shfl_xor(tb, 16, 64); tb += __shfl_xor(tb, 32, 64);
      t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
      if (g == 0) {
        o2[256 * C2 + r] = ta;
        o2[256 * C2 + 16 + r] = tb;
        o1[256 * C1 + r] = t1;
      }
    }
  }
  if (step_wd)                                               // grids below 242 workgroups (B < 121): the groups left over
    for (int s = blockIdx.x + gridDim.x; s < KSTEPS_DENSE; s += gridDim.x) {
      wd_step_load(u, g_wd, s, wst);
      wd_step_apply(u, s, wst);
    }
  GA3C_STAMP(7);
}

// ------------------------------------------------------------------ slab reduce (fixed order => reproducible)
// out_w[e] (e < nw) and out_b[e-nw] (nw <= e < stride) = sum_c part[c*stride + e], for two slab sets in one
// launch (conv1 and conv2 weight-gradient partials).  Block = 16 waves on the same 64 columns; wave w folds
// chunks w, w+16, ... (8 loads in flight), then wave 0 adds the 16 partial rows in order.
struct SlabSet { const float* part; int nchunks; int stride; int nw; float* out_w; float* out_b; int nblocks; int64_t off_w; int64_t off_b; };

template <bool UPD>
__global__ __launch_bounds__(1024) void slab_reduce_kernel(SlabSet s0, SlabSet s1, FusedUpd u) {
  __shared__ float sh[16][64];
  const bool first = (int)blockIdx.x < s0.nblocks;
  const SlabSet& ss = first ? s0 : s1;
  const int blk = first ? blockIdx.x : blockIdx.x - s0.nblocks;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int e = blk * 64 + lane;
  const bool ok = e < ss.stride;
  float a0 = 0.f, a1 = 0.f;
  for (int c0 = w; c0 < ss.nchunks; c0 += 128) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + 16 * i;
      t[i] = (ok && c < ss.nchunks) ? ss.part[(size_t)c * ss.stride + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; i += 2) { a0 += t[i]; a1 += t[i + 1]; }
  }
  sh[w][lane] = a0 + a1;
  __syncthreads();
  if (w == 0 && ok) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += sh[i][lane];
    if (e < ss.nw) ss.out_w[e] = tot; else ss.out_b[e - ss.nw] = tot;
    if (UPD) {    // the conv parameters are stepped where their gradient is completed (off_w / off_b: arena offsets)
      const int64_t i = e < ss.nw ? ss.off_w + e : ss.off_b + (e - ss.nw);
      const float tn = fused_rmsprop(u, i, tot);
      store_conv_packs(u.pk, i, tn);
    }
  }
}

// ------------------------------------------------------------------ gradient clipping (optional)
// tf.clip_by_average_norm per tensor: scale = clip / max(||g||_2 / n, clip).  One block per tensor.
struct TensorTable { int64_t off[11]; };
__global__ __launch_bounds__(256) void clip_scale_kernel(const float* __restrict__ grad, TensorTable tt, float clip,
                                                         float* __restrict__ scales) {
  __shared__ float sh[4];
  const int64_t lo = tt.off[blockIdx.x], hi = tt.off[blockIdx.x + 1];
  float s = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += grad[i] * grad[i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) scales[blockIdx.x] = clip / fmaxf(sqrtf(s) / (float)(hi - lo), clip);
}

// ------------------------------------------------------------------ RMSProp (TF-1.x ApplyRMSProp arithmetic)
// ms += (g*g - ms)*(1-rho); mom = mom*mu + (g*lr)/sqrt(eps+ms); theta_out = theta_in - mom.
// theta_in/theta_out are the two halves of the double-buffered weights.  Blocks [0,242) own one 16-row
// group of dense1/w each (thread n: rows 16s..16s+15 of column n), which is exactly one 64-byte row of the
// fragment-ordered copy pk_out that dense1_fwd reads; the remaining blocks sweep the rest of the arena.
template <bool CLIP, bool MOM>
__device__ __forceinline__ float rmsprop_one(int64_t i, const float* __restrict__ theta_in, float* __restrict__ theta_out,
                                             float* __restrict__ ms, float* __restrict__ mom,
                                             const float* __restrict__ grad, float lr, float one_minus_rho, float mu,
                                             float eps, float scale) {
  float g = grad[i];
  if (CLIP) g *= scale;
  float m = ms[i];
  m += (g * g - m) * one_minus_rho;
  ms[i] = m;
  float step = (g * lr) / sqrtf(eps + m);
  if (MOM) {
    step = mom[i] * mu + step;
    mom[i] = step;
  }
  const float tn = theta_in[i] - step;
  theta_out[i] = tn;
  return tn;
}

constexpr int RMS_WD_BLOCKS = KSTEPS_DENSE;   // 242
template <bool CLIP, bool MOM>
__global__ __launch_bounds__(256) void rmsprop_kernel(const float* __restrict__ theta_in, float* __restrict__ theta_out,
                                                      float* __restrict__ ms, float* __restrict__ mom,
                                                      const float* __restrict__ grad, int64_t n, float lr,
                                                      float one_minus_rho, float mu, float eps, TensorTable tt,
                                                      const float* __restrict__ scales, float* __restrict__ pk_out) {
  if (blockIdx.x < RMS_WD_BLOCKS) {
    const int s = blockIdx.x, col = threadIdx.x;
    const float sc = CLIP ? scales[4] : 1.0f;            // tensor 4 = dense1/w
    float v[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      v[kk] = rmsprop_one<CLIP, MOM>(OFF_WD + (int64_t)(s * 16 + kk) * HID + col, theta_in, theta_out, ms, mom, grad, lr,
                                     one_minus_rho, mu, eps, sc);
    float* dst = pk_out + ((size_t)s * HID + col) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<f32x4*>(dst + 4 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    return;
  }
  // everything except dense1/w: [0, OFF_WD) and [OFF_BD, n)
  const int64_t rest = OFF_WD + (n - OFF_BD);
  for (int64_t j = (int64_t)(blockIdx.x - RMS_WD_BLOCKS) * 256 + threadIdx.x; j < rest;
       j += (int64_t)(gridDim.x - RMS_WD_BLOCKS) * 256) {
    const int64_t i = j < OFF_WD ? j : j - OFF_WD + OFF_BD;
    float sc = 1.0f;
    if (CLIP) {
      int ti = 0;
#pragma unroll
      for (int k = 1; k < 10; ++k) ti += (i >= tt.off[k]) ? 1 : 0;
      sc = scales[ti];
    }
    const float tn = rmsprop_one<CLIP, MOM>(i, theta_in, theta_out, ms, mom, grad, lr, one_minus_rho, mu, eps, sc);
    store_conv_packs(pk_out, i, tn);
  }
}

}  // namespace ga3c
