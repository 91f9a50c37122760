// tools/kprobe.hip -- DIAGNOSTIC build of the kernels with in-kernel time stamps (development aid; not part of the
// product or the tests).  Prints, per kernel, where a workgroup's waves spend their cycles.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -DGA3C_STAMPS -I ga3c_amd/csrc -o tools/kprobe tools/kprobe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ga3c_kernels.hpp"
using namespace ga3c;

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128;
  hipStream_t st; CK(hipStreamCreate(&st));
  float *x, *w, *n1, *n2; uint8_t* xu8;
  CK(hipMalloc(&x, (size_t)B * XS * 4)); CK(hipMalloc(&xu8, (size_t)B * XS)); CK(hipMalloc(&w, 8 << 20));
  CK(hipMalloc(&n1, (size_t)B * N1S * 4)); CK(hipMalloc(&n2, (size_t)B * FLAT * 4));
  std::vector<float> hx((size_t)B * XS); std::vector<uint8_t> hk((size_t)B * XS);
  for (size_t i = 0; i < hx.size(); ++i) { hk[i] = (uint8_t)((i * 2654435761u) >> 24); hx[i] = (float)hk[i] / 128.f - 1.f; }
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(xu8, hk.data(), hk.size(), hipMemcpyHostToDevice));
  std::vector<float> hw(2 << 20);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) & 1023) / 8192.f - 0.06f;
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const int nwg = B * 2, NS = 8;
  unsigned long long* sb; CK(hipMalloc(&sb, (size_t)nwg * 16 * 16 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(ga3c_stamp_buf), &sb, sizeof sb));
  const size_t lds = CS_LDS_FLOATS * sizeof(float);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stack_fwd_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stack_fwd_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int u8 = 0; u8 < 2; ++u8) {
    for (int it = 0; it < 4; ++it) {
      CK(hipMemsetAsync(sb, 0, (size_t)nwg * 16 * 16 * 8, st));
      if (u8) hipLaunchKernelGGL((conv_stack_fwd_kernel<false, true>), dim3(nwg), dim3(1024), lds, st, (const void*)xu8, w, w + 4096, w + 8192, w + 20000, n1, n2, B, (const int64_t*)nullptr, SrcOffsets{}, (uint8_t*)nullptr);
      else hipLaunchKernelGGL((conv_stack_fwd_kernel<false, false>), dim3(nwg), dim3(1024), lds, st, (const void*)x, w, w + 4096, w + 8192, w + 20000, n1, n2, B, (const int64_t*)nullptr, SrcOffsets{}, (uint8_t*)nullptr);
      CK(hipStreamSynchronize(st));
    }
    std::vector<unsigned long long> h((size_t)nwg * 16 * 16);
    CK(hipMemcpy(h.data(), sb, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    std::vector<std::vector<double>> seg(NS), segmax(NS);
    std::vector<double> start, wgspan;
    for (int g = 0; g < nwg; ++g) {
      unsigned long long w0 = ~0ull, w1 = 0;
      std::vector<double> mx(NS, 0.0);
      for (int wv = 0; wv < 16; ++wv) {
        const unsigned long long* s = &h[((size_t)g * 16 + wv) * 16];
        if (s[0]) t0 = std::min(t0, s[0]); t1 = std::max(t1, s[7]); w0 = std::min(w0, s[0]); w1 = std::max(w1, s[7]);
        for (int k = 1; k < NS; ++k) { const double d = (double)(s[k] - s[k - 1]); if (wv == 0) seg[k].push_back(d); mx[k] = std::max(mx[k], d); }
      }
      for (int k = 1; k < NS; ++k) segmax[k].push_back(mx[k]);
      wgspan.push_back((double)(w1 - w0));
    }
    for (int g = 0; g < nwg; ++g) start.push_back((double)(h[(size_t)g * 256] - t0));
    std::sort(start.begin(), start.end());
    printf("conv_stack_fwd<%s> B=%d: kernel span %llu ticks; WG start spread: median %.0f last %.0f; WG span median %.0f\n",
           u8 ? "u8" : "f32", B, t1 - t0, med(start), start.back(), med(wgspan));
    const char* names[NS] = {"", "issue loads, zero n1", "W1 (u8: x too) -> LDS", "-", "barrier 1", "conv1 tile, W2 -> LDS", "barrier 2", "conv2 + stores"};
    for (int k = 1; k < NS; ++k) printf("  %-36s wave0 median %7.0f   slowest-wave median %7.0f ticks\n", names[k], med(seg[k]), med(segmax[k]));
    for (int wsel : {0, 8}) {                              // cumulative, from the workgroup's first stamp: an early and a late wave
      printf("  wave %d, ticks since the workgroup started:", wsel);
      for (int k = 0; k < NS; ++k) {
        std::vector<double> c;
        for (int g = 0; g < nwg; ++g) {
          unsigned long long w0 = ~0ull;
          for (int wv = 0; wv < 16; ++wv) w0 = std::min(w0, h[((size_t)g * 16 + wv) * 16]);
          c.push_back((double)(h[((size_t)g * 16 + wsel) * 16 + k] - w0));
        }
        printf(" [%d] %.0f", k, med(c));
      }
      printf("\n");
    }
  }
  // ---- conv_bwd_kernel (f32): one sample per workgroup at B = 128
  {
    float *dn2, *dn1, *slab2, *slab1, *pk;
    CK(hipMalloc(&dn2, (size_t)B * FLAT * 4)); CK(hipMalloc(&dn1, (size_t)B * N1S * 4));
    CK(hipMalloc(&slab2, (size_t)256 * SLAB2 * 4)); CK(hipMalloc(&slab1, (size_t)512 * SLAB1 * 4)); CK(hipMalloc(&pk, 8192 * 4));
    CK(hipMemcpy(dn2, hx.data(), (size_t)B * FLAT * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(n1, hx.data() + 12345, (size_t)B * N1S * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(pk, hw.data(), 8192 * 4, hipMemcpyHostToDevice));
    const size_t cl = CB_LDS_FLOATS * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cl));
    const int grid = 2 * B;
    for (int it = 0; it < 4; ++it) {
      CK(hipMemsetAsync(sb, 0, (size_t)nwg * 16 * 16 * 8, st));
      hipLaunchKernelGGL(conv_bwd_kernel<false>, dim3(grid), dim3(1024), cl, st, (const void*)x, n1, dn2, pk, dn1, slab2, slab1, B, (const float*)nullptr, FusedUpd{});
      CK(hipStreamSynchronize(st));
    }
    std::vector<unsigned long long> h((size_t)nwg * 16 * 16);
    CK(hipMemcpy(h.data(), sb, h.size() * 8, hipMemcpyDeviceToHost));
    const char* names[8] = {"", "W2 + zeroing + barrier", "stage sample + barrier", "phase 1 (dn1)", "phase 2 (dW2)", "barrier", "phase 3 (dW1, bands)", "fold + slab stores"};
    std::vector<std::vector<double>> seg(8), segmax(8);
    for (int gq = 0; gq < grid; ++gq) {
      std::vector<double> mx(8, 0.0);
      for (int wv = 0; wv < 16; ++wv) {
        const unsigned long long* s = &h[((size_t)gq * 16 + wv) * 16];
        for (int k = 1; k < 8; ++k) { const double d = (double)(s[k] - s[k - 1]); if (wv == 0) seg[k].push_back(d); mx[k] = std::max(mx[k], d); }
      }
      for (int k = 1; k < 8; ++k) segmax[k].push_back(mx[k]);
    }
    printf("conv_bwd<f32> B=%d\n", B);
    for (int k = 1; k < 8; ++k) printf("  %-24s wave0 median %7.0f   slowest-wave median %7.0f ticks\n", names[k], med(seg[k]), med(segmax[k]));
  }
  return 0;
}
