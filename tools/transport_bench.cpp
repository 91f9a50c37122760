// tools/transport_bench.cpp -- what one prediction costs the HOST, with the GPU replaced by a timer.
// N agent threads speak the agent side of include/ga3c_host.h (state into the slot, submit, sleep on the slot's futex), K
// predictor threads run the product's native serve loop (ga3c_pq_serve_pipelined) over begin / end callbacks that stand
// in for the network: `end` returns `latency_us` after `begin` (spinning or sleeping, as the HIP runtime's wait would).
// Prints predictions/s, the mean batch and the CPU time per prediction split into agents / predictor loops / the rest
// (answering helpers).  Runs without a GPU: a development aid for the transport, not part of the product.
//   g++ -O2 -std=c++17 -pthread -I include -o tools/transport_bench tools/transport_bench.cpp -L ga3c_amd -lga3c_host -Wl,-rpath,$PWD/ga3c_amd
//   tools/transport_bench <agents> <predictors> <seconds> <latency_us> [spin|sleep] [agent_work_us]
#include <sys/resource.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ga3c_host.h"

static int64_t now_ns() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}
static int64_t thread_cpu_ns() {
  struct timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}

struct FakeNet {
  int latency_us = 60;
  bool spin = true;
  int num_actions = 6;
  int64_t t_begin[4] = {0, 0, 0, 0};
  int next = 0;
};

static int fake_begin(void* net, const int64_t*, int32_t, int32_t, int32_t* ticket) {
  FakeNet* f = static_cast<FakeNet*>(net);
  *ticket = f->next;
  f->t_begin[f->next] = now_ns();
  f->next = (f->next + 1) & 3;
  return 0;
}
static int fake_end(void* net, int32_t ticket, int32_t batch, float* p, float* v) {
  FakeNet* f = static_cast<FakeNet*>(net);
  const int64_t until = f->t_begin[ticket] + (int64_t)f->latency_us * 1000;
  if (f->spin) {
    while (now_ns() < until) __builtin_ia32_pause();
  } else {
    const int64_t left = until - now_ns();
    if (left > 0) {
      struct timespec ts = {0, (long)left};
      nanosleep(&ts, nullptr);
    }
  }
  for (int i = 0; i < batch; ++i) {
    for (int a = 0; a < f->num_actions; ++a) p[(size_t)i * f->num_actions + a] = 1.f / f->num_actions;
    v[i] = 0.f;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s <agents> <predictors> <seconds> <latency_us> [spin|sleep] [agent_work_us]\n", argv[0]); return 2; }
  const int n = atoi(argv[1]), np = atoi(argv[2]);
  const double seconds = atof(argv[3]);
  const int latency = atoi(argv[4]);
  const bool spin = argc < 6 || std::string(argv[5]) == "spin";
  const int work_us = argc > 6 ? atoi(argv[6]) : 0;
  ga3c_shm_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.max_agents = n; cfg.num_actions = 6; cfg.state_bytes = 84 * 84 * 4; cfg.train_slots = 100; cfg.train_rows = 6;
  char name[64];
  snprintf(name, sizeof name, "/ga3c_tb_%d", (int)getpid());
  ga3c_shm* shm = nullptr;
  if (ga3c_shm_create(name, &cfg, &shm) != 0) { fprintf(stderr, "create: %s\n", ga3c_host_last_error()); return 1; }
  const size_t sb = (size_t)cfg.state_bytes;
  std::vector<unsigned char> pool(64 * sb, 7);
  std::atomic<bool> stop{false};
  std::atomic<long long> steps{0}, cpu_agents{0}, cpu_pred{0}, batches{0}, served{0}, ns_resp{0}, ns_pred{0}, ns_pop{0};
  std::vector<std::thread> th;
  for (int id = 0; id < n; ++id) {
    th.emplace_back([&, id] {
      std::vector<float> p(cfg.num_actions);
      float v;
      int k = id;
      unsigned char* slot = static_cast<unsigned char*>(ga3c_pq_state_ptr(shm, id));
      long long mine = 0;
      while (!stop.load(std::memory_order_relaxed)) {
        k = (k + 1) & 63;
        memcpy(slot, &pool[(size_t)k * sb], sb);
        if (work_us > 0) { const int64_t u = now_ns() + (int64_t)work_us * 1000; while (now_ns() < u) __builtin_ia32_pause(); }
        if (ga3c_pq_submit(shm, id) != 0) break;
        int rc;
        while ((rc = ga3c_pq_wait(shm, id, p.data(), &v, 200)) != 0 && !stop.load(std::memory_order_relaxed))
          if (rc != GA3C_H_ETIMEOUT) break;
        if (rc != 0) break;
        ++mine;
      }
      steps.fetch_add(mine);
      cpu_agents.fetch_add(thread_cpu_ns());
    });
  }
  std::vector<std::thread> pt;
  std::vector<FakeNet> nets((size_t)np);
  for (int k = 0; k < np; ++k) {
    nets[k].latency_us = latency; nets[k].spin = spin;
    pt.emplace_back([&, k] {
      ga3c_serve_stats st;
      memset(&st, 0, sizeof st);
      while (!stop.load(std::memory_order_relaxed)) {
        const int rc = ga3c_pq_serve_pipelined(shm, fake_begin, fake_end, &nets[k], 1, 128, 50, &st);
        if (rc != 0) break;
      }
      batches.fetch_add(st.batches); served.fetch_add(st.served);
      ns_resp.fetch_add(st.ns_respond); ns_pred.fetch_add(st.ns_predict); ns_pop.fetch_add(st.ns_pop);
      cpu_pred.fetch_add(thread_cpu_ns());
    });
  }
  const int64_t t0 = now_ns();
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& t : pt) t.join();
  ga3c_shm_shutdown(shm);
  for (auto& t : th) t.join();
  const double dt = (now_ns() - t0) * 1e-9;
  struct rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  const double cpu_all = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 + ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
  const double sys_all = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
  const double pred = (double)served.load();
  const double b = (double)(batches.load() > 0 ? batches.load() : 1);
  printf("{\"agents\": %d, \"predictors\": %d, \"latency_us\": %d, \"wait\": \"%s\", \"predictions_per_sec\": %.0f, \"mean_batch\": %.1f, "
         "\"cores_used\": %.2f, \"sys_share\": %.2f, \"cpu_us_per_prediction\": {\"total\": %.2f, \"agents\": %.2f, \"predictor_loops\": %.2f, \"rest\": %.2f}, "
         "\"loop_us_per_batch\": {\"pop\": %.1f, \"predict\": %.1f, \"respond\": %.1f}, \"voluntary_switches\": %ld, \"involuntary_switches\": %ld}\n",
         n, np, latency, spin ? "spin" : "sleep", pred / dt, pred / b, cpu_all / dt, sys_all / cpu_all, cpu_all / pred * 1e6,
         cpu_agents.load() * 1e-3 / pred, cpu_pred.load() * 1e-3 / pred, (cpu_all * 1e9 - cpu_agents.load() - cpu_pred.load()) * 1e-3 / pred,
         ns_pop.load() * 1e-3 / b, ns_pred.load() * 1e-3 / b, ns_resp.load() * 1e-3 / b, ru.ru_nvcsw, ru.ru_nivcsw);
  ga3c_shm_close(shm, 1);
  return 0;
}
