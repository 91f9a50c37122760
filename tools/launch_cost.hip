// tools/launch_cost.hip -- host cost of a kernel launch from 1..6 threads at once (own stream each), hipLaunchKernelGGL against
// hipModuleLaunchKernel on a function handle resolved once (development aid: is the launch path what the engine's predictor and
// trainer threads contend for?)
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/launch_cost tools/launch_cost.hip
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1);} } while (0)
struct Args { float* p; int n; float a, b; long pad[8]; };
__global__ void small_kernel(Args a) { if (threadIdx.x == 9999) a.p[0] = a.a + a.b + a.n; }
int main() {
  float* buf; CK(hipMalloc(&buf, 4096));
  hipFunction_t fn; CK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(&small_kernel)));
  for (int mode = 0; mode < 2; ++mode)
    for (int nt : {1, 2, 4, 6}) {
      std::atomic<int> go{0};
      std::vector<double> us(nt);
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
          hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
          Args a{buf, 3, 1.f, 2.f, {}};
          size_t sz = sizeof a;
          void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
          for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(small_kernel, dim3(32), dim3(256), 0, st, a);
          CK(hipStreamSynchronize(st));
          go.fetch_add(1);
          while (go.load() < nt) {}
          const auto t0 = std::chrono::steady_clock::now();
          const int N = 20000;
          for (int i = 0; i < N; ++i) {
            if (mode == 0) hipLaunchKernelGGL(small_kernel, dim3(32), dim3(256), 0, st, a);
            else CK(hipModuleLaunchKernel(fn, 32, 1, 1, 256, 1, 1, 0, st, nullptr, cfg));
            if ((i & 63) == 63) CK(hipStreamSynchronize(st));          // keep the queue from filling up
          }
          CK(hipStreamSynchronize(st));
          us[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
          CK(hipStreamDestroy(st));
        });
      for (auto& x : th) x.join();
      double m = 0; for (double v : us) m += v / nt;
      printf("%-24s %d thread(s): %.2f us per launch (incl. a sync every 64)\n", mode ? "hipModuleLaunchKernel" : "hipLaunchKernelGGL", nt, m);
    }
  return 0;
}
