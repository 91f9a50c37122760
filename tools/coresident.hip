// tools/coresident.hip -- can a small LDS-free kernel on a second stream run BESIDE the one-workgroup-per-CU fused conv
// kernels (150-158 KB of LDS, 16 waves per CU), and what does it cost them?  (development aid; the question behind the
// tail of a 129..136-row train step: DESIGN.md section 7, "the 128-row cliff")
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -DGA3C_STAMPS -DGA3C_STAMPS_REALTIME -I ga3c_amd/csrc -o tools/coresident tools/coresident.hip
// Times are in-kernel stamps of the 100 MHz counter every XCD shares (10 ns), not events: an event pair costs ~3 us itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "ga3c_kernels.hpp"
using namespace ga3c;

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)

// LDS-free stand-in for a tail kernel: per lane `nld` 16-byte loads (L2-resident source) in batches of 8, then `nmf` MFMAs
__device__ __forceinline__ unsigned long long realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
__global__ __launch_bounds__(256, 2) void tail_proxy(const float* __restrict__ src, float* __restrict__ dst, int nld, int nmf,
                                                     unsigned long long* __restrict__ tt) {
  const unsigned long long t0 = realtime();
  const int gid = blockIdx.x * 256 + threadIdx.x;
  f32x4 acc0 = zero4(), acc1 = zero4();
  const float* p = src + (size_t)(gid & 0x3fff) * 4;         // 4 MB of source: L2-resident after the first pass
  for (int i = 0; i < nld; i += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld4(p + (size_t)((i + j) & 15) * 16384 * 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc0 = mfma(v[j][0], v[j][1], acc0);
      acc1 = mfma(v[j][2], v[j][3], acc1);
    }
  }
  float a = 0.37f + threadIdx.x * 1e-3f;
  for (int i = 0; i < nmf; i += 2) { acc0 = mfma(a, acc1[0], acc0); acc1 = mfma(acc0[1], a, acc1); }
  dst[gid] = acc0[0] + acc1[1] + acc0[2] + acc1[3];
  if (threadIdx.x == 0) { tt[2 * blockIdx.x] = t0; tt[2 * blockIdx.x + 1] = realtime(); }
}

static float med(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  const int B = 128;
  const int nb = argc > 1 ? atoi(argv[1]) : 128;           // proxy workgroups
  const int nld = argc > 2 ? atoi(argv[2]) : 64;
  const int nmf = argc > 3 ? atoi(argv[3]) : 64;
  const int chain = argc > 4 ? atoi(argv[4]) : 2;          // dependent proxies per "tail"
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t sa, st, sn;
  CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
  float *x, *pk, *th, *n1, *n2, *dn2, *slab1, *slab2, *src, *dst;
  CK(hipMalloc(&x, (size_t)B * XS * 4)); CK(hipMalloc(&pk, (size_t)PK_FLOATS * 4)); CK(hipMalloc(&th, 8 << 20));
  CK(hipMalloc(&n1, (size_t)B * N1S * 4)); CK(hipMalloc(&n2, (size_t)B * FLAT * 4)); CK(hipMalloc(&dn2, (size_t)B * FLAT * 4));
  CK(hipMalloc(&slab1, (size_t)512 * SLAB1 * 4)); CK(hipMalloc(&slab2, (size_t)256 * SLAB2 * 4));
  CK(hipMalloc(&src, (size_t)64 * 65536 * 16)); CK(hipMalloc(&dst, (size_t)4096 * 256 * 4));
  std::vector<float> h((size_t)B * XS);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 24) / 128.f - 1.f;
  CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> hw((size_t)PK_FLOATS);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) & 1023) / 8192.f - 0.06f;
  CK(hipMemcpy(pk, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(th, hw.data(), 1 << 20, hipMemcpyHostToDevice));
  CK(hipMemcpy(dn2, h.data(), (size_t)B * FLAT * 4, hipMemcpyHostToDevice));
  CK(hipMemset(src, 0, (size_t)64 * 65536 * 16));
  const size_t lds_f = CS_LDS_FLOATS * sizeof(float), lds_b = CB_LDS_FLOATS * sizeof(float);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stack_fwd_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
  auto fwd = [&](hipStream_t s) {
    hipLaunchKernelGGL((conv_stack_fwd_kernel<true, false>), dim3(B * 2), dim3(1024), lds_f, s, (const void*)x, pk + PK_W1F, th + OFF_B1,
                       pk + PK_W2F, th + OFF_B2, n1, n2, B, (const int64_t*)nullptr, SrcOffsets{}, (uint8_t*)nullptr); };
  auto bwd = [&](hipStream_t s) {
    hipLaunchKernelGGL(conv_bwd_kernel<false>, dim3(B * 2), dim3(1024), lds_b, s, (const void*)x, n1, dn2, pk + PK_W2DX, (float*)nullptr,
                       slab2, slab1, B, (const float*)nullptr, FusedUpd{}); };
  hipEvent_t e0; CK(hipEventCreate(&e0));
  unsigned long long *sb, *tt;                              // main kernel's stamps [wg][16 waves][16], proxies' [chain][wg][2]
  CK(hipMalloc(&sb, (size_t)256 * 16 * 16 * 8)); CK(hipMalloc(&tt, (size_t)8 * 4096 * 2 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ga3c_stamp_buf), &sb, sizeof sb));
  auto tail = [&](hipStream_t s) {
    for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(tail_proxy, dim3(nb), dim3(256), 0, s, src, dst, nld, nmf, tt + (size_t)c * 4096 * 2); };
  std::vector<unsigned long long> hs((size_t)256 * 256), ht((size_t)8 * 4096 * 2);
  // span of the main kernel (first stamp of any wave .. last stamp of any wave) and of each proxy, relative to `origin`
  auto read_main = [&](int last, double& b, double& e) {
    CK(hipMemcpy(hs.data(), sb, hs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < 256 * 16; ++w) { lo = std::min(lo, hs[(size_t)w * 16]); hi = std::max(hi, hs[(size_t)w * 16 + last]); }
    b = (double)lo; e = (double)hi;
  };
  auto read_tail = [&](int c, double& b, double& e) {
    CK(hipMemcpy(ht.data(), tt, ht.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < nb; ++w) { lo = std::min(lo, ht[((size_t)c * 4096 + w) * 2]); hi = std::max(hi, ht[((size_t)c * 4096 + w) * 2 + 1]); }
    b = (double)lo; e = (double)hi;
  };
  printf("proxy: %d workgroups x 256, %d loads + %d MFMAs per lane, chain of %d (times in us from the first kernel's first wave)\n", nb, nld, 2 * nld + nmf, chain);
  for (int which = 0; which < 2; ++which) {
    const char* name = which ? "conv_bwd" : "conv_stack_fwd<train>";
    const int last = 7;                          // the kernel's last stamp index
    auto maink = [&](hipStream_t s) { if (which) bwd(s); else fwd(s); };
    for (int tail_stream = 0; tail_stream < 3; ++tail_stream) {
      hipStream_t ts = tail_stream == 2 ? sa : tail_stream ? sn : st;
      std::vector<float> m_alone, m_len, t_first, t_last, t_alone;
      for (int it = 0; it < 30; ++it) {
        double b, e, tb, te;
        maink(sa); CK(hipStreamSynchronize(sa)); read_main(last, b, e);
        if (it >= 5) m_alone.push_back((float)((e - b) / 100.0));
        tail(ts); CK(hipStreamSynchronize(ts)); read_tail(0, tb, te); { double b2, e2; read_tail(chain - 1, b2, e2); te = e2; }
        if (it >= 5) t_alone.push_back((float)((te - tb) / 100.0));
        CK(hipEventRecord(e0, sa)); if (ts != sa) CK(hipStreamWaitEvent(ts, e0, 0));
        maink(sa); tail(ts);
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(ts));
        read_main(last, b, e); read_tail(0, tb, te);
        double org = std::min(b, tb), b2, e2; read_tail(chain - 1, b2, e2);
        if (it >= 5) { m_len.push_back((float)((e - org) / 100.0)); t_first.push_back((float)((tb - org) / 100.0)); t_last.push_back((float)((e2 - org) / 100.0)); }
      }
      printf("%-22s tail on %s: main alone %.2f, tail alone %.2f; together: main ends at %.2f, tail starts at %.2f, ends at %.2f\n",
             name, tail_stream == 2 ? "the SAME stream       " : tail_stream ? "a normal-priority stream" : "a 2nd high-prio stream",
             med(m_alone), med(t_alone), med(m_len), med(t_first), med(t_last));
    }
  }
  return 0;
}
