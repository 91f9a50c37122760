// tools/kbench.hip -- kernel micro-benchmarks (development aid; not part of the product or the tests).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I ga3c_amd/csrc -o tools/kbench tools/kbench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "ga3c_kernels.hpp"
using namespace ga3c;

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void mfma_loop(float* out, int n, float seed) {
  f32x4 a0 = zero4(), a1 = zero4();
  float x = seed + threadIdx.x * 1e-3f, y = seed * 0.5f;
  for (int i = 0; i < n; i += 2) { a0 = mfma(x, y, a0); a1 = mfma(y, x, a1); }
  if (a0[0] + a1[1] == 12345.f) out[threadIdx.x] = a0[0];
}
__global__ void empty_kernel(float* p) { if (threadIdx.x == 9999) p[0] = 1.f; }

template <typename F>
float time_us(F launch, hipStream_t st, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double tot = 0;
  for (int i = 0; i < iters + 3; ++i) {
    launch(e0, e1);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 3) tot += ms;
  }
  return (float)(tot / iters * 1e3);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128;
  hipStream_t st; CK(hipStreamCreate(&st));
  float *x, *w, *n1;
  CK(hipMalloc(&x, (size_t)B * XS * 4)); CK(hipMalloc(&w, 8 << 20)); CK(hipMalloc(&n1, (size_t)B * N1S * 4));
  std::vector<float> hx((size_t)B * XS);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 24) / 128.f - 1.f;
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> hw(2 << 20);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) & 1023) / 8192.f - 0.06f;
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  for (int n : {128, 512, 2048, 8192, 32768}) {
    float us = time_us([&](hipEvent_t a, hipEvent_t b) {
      hipExtLaunchKernelGGL(mfma_loop, dim3(256), dim3(256), 0, st, a, b, 0, n1, n, 0.37f); }, st, 10);
    printf("mfma_loop 1 wave/SIMD n=%6d: %.2f us (%.1f cycles@2.4GHz per MFMA incl. floor)\n", n, us, us * 2400.0 / n);
  }
  auto chain = [&](const char* name, auto launch, int n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s back-to-back x%d: %.2f us each\n", name, n, ms / n * 1e3);
  };
  float *n2, *pk, *part;
  CK(hipMalloc(&n2, (size_t)B * FLAT * 4)); CK(hipMalloc(&pk, (size_t)FLAT * HID * 4)); CK(hipMalloc(&part, (size_t)22 * B * HID * 4));
  CK(hipMemset(pk, 0, (size_t)FLAT * HID * 4));
  printf("B=%d ideal MFMA us: conv1 %.2f conv2 %.2f dense1 %.2f\n", B, 3612672.0 * B / 157.3e12 * 1e6, 1982464.0 * B / 157.3e12 * 1e6, 1982464.0 * B / 157.3e12 * 1e6);
  for (int nb : {64, 256, 512, 896, 2048})
    printf("empty kernel %4d blocks single: %.2f us\n", nb, time_us([&](hipEvent_t a, hipEvent_t b) {
      hipExtLaunchKernelGGL(empty_kernel, dim3(nb), dim3(256), 0, st, a, b, 0, n1); }, st, 20));
  chain("empty 896 blocks", [&]() { hipLaunchKernelGGL(empty_kernel, dim3(896), dim3(256), 0, st, n1); }, 200);
  printf("conv1 single: %.2f us\n", time_us([&](hipEvent_t a, hipEvent_t b) {
    hipExtLaunchKernelGGL(conv1_fwd_kernel, dim3(B * 7), dim3(256), 0, st, a, b, 0, x, w, w + 4096, n1, B); }, st, 20));
  chain("conv1", [&]() { hipLaunchKernelGGL(conv1_fwd_kernel, dim3(B * 7), dim3(256), 0, st, x, w, w + 4096, n1, B); }, 200);
  printf("conv2 single: %.2f us\n", time_us([&](hipEvent_t a, hipEvent_t b) {
    hipExtLaunchKernelGGL(conv2_fwd_kernel, dim3(B * 2), dim3(256), 0, st, a, b, 0, n1, w, w + 8192, n2, B); }, st, 20));
  chain("conv2", [&]() { hipLaunchKernelGGL(conv2_fwd_kernel, dim3(B * 2), dim3(256), 0, st, n1, w, w + 8192, n2, B); }, 200);
  for (int mt : {1, 2, 4}) {
    for (int ks : {11, 22}) {
      const int nb = dense1_fwd_blocks(B, ks, mt);
      auto l1 = [&](hipEvent_t a, hipEvent_t b) {
        if (mt == 1) hipExtLaunchKernelGGL(dense1_fwd_kernel<1>, dim3(nb), dim3(256), 0, st, a, b, 0, n2, pk, part, B, ks, 242 / ks);
        else if (mt == 2) hipExtLaunchKernelGGL(dense1_fwd_kernel<2>, dim3(nb), dim3(256), 0, st, a, b, 0, n2, pk, part, B, ks, 242 / ks);
        else hipExtLaunchKernelGGL(dense1_fwd_kernel<4>, dim3(nb), dim3(256), 0, st, a, b, 0, n2, pk, part, B, ks, 242 / ks); };
      printf("dense1_fwd MT=%d KS=%d blocks=%d: %.2f us\n", mt, ks, nb, time_us(l1, st, 20));
    }
  }
  return 0;
}
