// tools/qmap.hip -- which HIP streams share a hardware queue, and how many queues the chip serves at a time.
// Development aid behind profiles/README.md ("Hardware queues"); not part of the product.
//   hipcc -O2 --offload-arch=gfx950 -o tools/qmap tools/qmap.hip && GPU_MAX_HW_QUEUES=4 ./tools/qmap
// 1. pairs: a 300-us spin kernel (ONE workgroup) on stream i, then a trivial kernel on stream j: j finishing after i means
//    the two streams are multiplexed onto one hardware queue (in-order), j finishing at once means they are not.
// 2. width: the spin kernel on n streams at once, one workgroup each: wall time / 300 us = how many turns the queues took.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin_kernel(long long ticks, int* sink) {   // s_memrealtime: 100 MHz
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (sink && ticks < 0) *sink = 1;
}
__global__ void tiny_kernel(int* sink) { if (sink && threadIdx.x > 1024) *sink = 1; }

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 8;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("GPU_MAX_HW_QUEUES=%s, %d normal streams + 1 high-priority stream (priority range %d..%d)\n",
         getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "default", S, lo, hi);
  std::vector<hipStream_t> st(S + 1);
  if (argc > 2) {      // any second argument: use the null stream first, as a program that calls hipMemset / hipMemcpy does
    int* p = nullptr;
    CK(hipMalloc((void**)&p, 64));
    CK(hipMemset(p, 0, 64));
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, nullptr, nullptr);
    CK(hipDeviceSynchronize());
    printf("(the null stream has been used before the streams are created: pairs row/column N = the null stream)\n");
  }
  for (int i = 0; i < S; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&st[S], hipStreamNonBlocking, hi));
  for (auto s : st) { hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, nullptr); }
  CK(hipDeviceSynchronize());
  const long long SPIN = 30000;   // 300 us
  printf("pairs: '#' = stream j (column) waited for the spin kernel on stream i (row): same hardware queue\n     ");
  for (int j = 0; j <= S; ++j) printf("%3d", j);
  printf("   (stream %d = high priority)\n", S);
  for (int i = 0; i <= S; ++i) {
    printf("  %2d ", i);
    for (int j = 0; j <= S; ++j) {
      if (i == j) { printf("  ."); continue; }
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[i], SPIN, nullptr);
      const double t0 = now_us();
      hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st[j], nullptr);
      CK(hipStreamSynchronize(st[j]));
      const double dt = now_us() - t0;
      CK(hipDeviceSynchronize());
      printf("  %c", dt > 150.0 ? '#' : '-');
    }
    printf("\n");
  }
  if (argc > 2) {
    printf("   N ");
    for (int j = 0; j <= S; ++j) {
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, nullptr, SPIN, nullptr);
      const double t0 = now_us();
      hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st[j], nullptr);
      CK(hipStreamSynchronize(st[j]));
      const double dt = now_us() - t0;
      CK(hipDeviceSynchronize());
      printf("  %c", dt > 150.0 ? '#' : '-');
    }
    printf("\n");
  }
  printf("width: n streams each running one 300-us single-workgroup kernel, wall time of the lot\n");
  for (int n = 1; n <= S; ++n) {
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipDeviceSynchronize());
      const double t0 = now_us();
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[i], SPIN, nullptr);
      CK(hipDeviceSynchronize());
      const double dt = now_us() - t0;
      if (dt < best) best = dt;
    }
    printf("  n = %d: %7.1f us  (%.2f turns)\n", n, best, best / 300.0);
  }
  // chains of short dependent kernels, the engine's regime: every stream runs 200 back-to-back 5-us kernels
  printf("chains: n streams each running 200 back-to-back 5-us single-workgroup kernels: us per kernel per stream\n");
  for (int n = 1; n <= S; ++n) {
    CK(hipDeviceSynchronize());
    const double t0 = now_us();
    for (int k = 0; k < 200; ++k)
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[i], 500LL, nullptr);
    CK(hipDeviceSynchronize());
    printf("  n = %d: %6.2f us per kernel per stream, %6.2f M kernels/s in all\n", n, (now_us() - t0) / 200.0, n * 200.0 / (now_us() - t0));
  }
  // hipExtAnyOrderLaunch: two 300-us kernels on ONE stream, the second launched without the ordering barrier
  {
    double best[2] = {1e30, 1e30};
    for (int flag = 0; flag < 2; ++flag)
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now_us();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], SPIN, nullptr);
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, SPIN, (int*)nullptr);
        CK(hipStreamSynchronize(st[0]));
        const double dt = now_us() - t0;
        if (dt < best[flag]) best[flag] = dt;
      }
    printf("any-order launch: two 300-us kernels on one stream: %.1f us in order, %.1f us with hipExtAnyOrderLaunch\n", best[0], best[1]);
  }
  return 0;
}
