// tests/native/native_agents.cpp -- the engine's ceiling when actors cost (almost) nothing; also the 256-agent scale test
// of BASELINE configs[3] / [4] (tests/test_gpu_engine_e2e.py) and the actor side of tools/engine_ceiling.py.
// N threads act as agents 0..N-1 of a running Server through the C ABI of include/ga3c_host.h, exactly as ProcessAgent does
// (ProcessAgent.py:102-107,117-162,175 of the reference): write a fresh uint8 state into the slot, submit, sleep on the
// slot's futex for (p, v), draw the action from p, and every TIME_MAX steps ship a rollout (states, returns, actions) to the
// training queue.  The "emulator" is a memcpy out of a pool of random frames, so what is measured is the transport, the
// batching predictors / trainers and the GPU -- not Python.  Not part of the product.
//   g++ -O2 -std=c++17 -pthread -I include -o tools/native_agents tests/native/native_agents.cpp -L ga3c_amd -lga3c_host -Wl,-rpath,$PWD/ga3c_amd
//   tools/native_agents <segment name> <agents> <seconds> <train 0|1>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "ga3c_host.h"

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s <segment> <agents> <seconds> <train 0|1>\n", argv[0]); return 2; }
  const char* name = argv[1];
  const int n = atoi(argv[2]);
  const double seconds = atof(argv[3]);
  const bool train = atoi(argv[4]) != 0;
  ga3c_shm* shm = nullptr;
  ga3c_host_signal_hold(0);                     // the agents here are threads: their signal masks share one kernel lock
  if (ga3c_shm_attach(name, &shm) != 0) { fprintf(stderr, "attach: %s\n", ga3c_host_last_error()); return 1; }
  ga3c_shm_config cfg;
  ga3c_shm_get_config(shm, &cfg);
  if (n > cfg.max_agents) { fprintf(stderr, "segment does not fit (%d agents max)\n", cfg.max_agents); return 1; }
  // rollout_row_bytes == 16: the frame queue lives on the device (Config.FRONTEND = 'device'): the slot carries the newest
  // frame / plane, requests carry flags, rollout rows name their state as (plane sequence number, agent id)
  // ... or, with "cache" as the fifth argument, the engine's state cache is in use (Config.STATE_CACHE): the slot carries the
  // whole state as ever, rollout rows name it as (request number, agent id)
  const bool cache = argc > 5 && std::string(argv[5]) == "cache";
  if (cache && cfg.rollout_row_bytes != 16) { fprintf(stderr, "\"cache\" needs 16-byte rollout rows\n"); return 1; }
  const bool device = cfg.rollout_row_bytes == 16 && !cache;
  const size_t sb = (size_t)cfg.state_bytes;
  std::vector<unsigned char> pool(64 * sb);
  std::mt19937_64 rng(12345);
  for (size_t i = 0; i + 8 <= pool.size(); i += 8) { const unsigned long long r = rng(); memcpy(&pool[i], &r, 8); }
  std::atomic<bool> stop{false};
  std::atomic<long long> steps{0}, rollouts{0}, ns_copy{0}, ns_submit{0}, ns_wait{0};
  std::vector<std::thread> th;
  for (int id = 0; id < n; ++id) {
    th.emplace_back([&, id] {
      std::mt19937 r(1000 + id);
      std::vector<float> p(cfg.num_actions);
      std::vector<int> acts(cfg.train_rows);
      std::vector<int> frames(cfg.train_rows);
      std::vector<long long> seqs(cfg.train_rows);
      long long pushed = 0;
      float v = 0.f;
      int t = 0, k = id;
      unsigned char* slot = static_cast<unsigned char*>(ga3c_pq_state_ptr(shm, id));
      long long mine = 0;
      while (!stop.load(std::memory_order_relaxed)) {
        k = (k + 1) & 63;
        const auto c0 = std::chrono::steady_clock::now();
        memcpy(slot, &pool[(size_t)k * sb], sb);               // the emulator's next state
        const auto c1 = std::chrono::steady_clock::now();
        if (device) {
          const uint32_t fl = (pushed == 0 ? GA3C_REQ_RESET : 0u) | (pushed < 3 ? GA3C_REQ_NO_PREDICT : 0u);
          if (ga3c_pq_submit_flags(shm, id, fl) != 0) break;
          ++pushed;
        } else if (ga3c_pq_submit(shm, id) != 0) break;
        const auto c2 = std::chrono::steady_clock::now();
        int rc;
        while ((rc = ga3c_pq_wait(shm, id, p.data(), &v, 200)) != 0 && !stop.load(std::memory_order_relaxed))
          if (rc != GA3C_H_ETIMEOUT) return;
        if (rc != 0) break;
        const auto c3 = std::chrono::steady_clock::now();
        if ((mine & 15) == 0) {                                 // wall time of the three parts of a step, sampled
          ns_copy.fetch_add((c1 - c0).count(), std::memory_order_relaxed);
          ns_submit.fetch_add((c2 - c1).count(), std::memory_order_relaxed);
          ns_wait.fetch_add((c3 - c2).count(), std::memory_order_relaxed);
        }
        if (device && pushed < 4) continue;                     // the queue was still filling: no prediction came back
        float u = std::generate_canonical<float, 24>(r), c = 0.f;
        int a = cfg.num_actions - 1;
        for (int i = 0; i < cfg.num_actions; ++i) { c += p[i]; if (u < c) { a = i; break; } }
        acts[t] = a; frames[t] = k; seqs[t] = pushed - 1;
        if (cache) { int64_t rq = 0; ga3c_pq_request_seq(shm, id, &rq); seqs[t] = rq; }
        ++mine;
        if (++t == cfg.train_rows - 1) {                        // TIME_MAX steps: ship the rollout
          if (train) {
            int s;
            while ((s = ga3c_tq_acquire(shm, 200)) < 0 && !stop.load(std::memory_order_relaxed))
              if (s != GA3C_H_ETIMEOUT) return;
            if (s < 0) break;
            unsigned char* st = static_cast<unsigned char*>(ga3c_tq_states(shm, s));
            float* ret = ga3c_tq_returns(shm, s);
            int32_t* ac = ga3c_tq_actions(shm, s);
            for (int i = 0; i < t; ++i) {
              if (device || cache) { memcpy(st + (size_t)i * 16, &seqs[i], 8); const int32_t me = id; memcpy(st + (size_t)i * 16 + 8, &me, 4); }
              else memcpy(st + (size_t)i * sb, &pool[(size_t)frames[i] * sb], sb);
              ret[i] = 0.01f * (float)(i - 2); ac[i] = acts[i];
            }
            ga3c_tq_commit(shm, s, t);
            rollouts.fetch_add(1, std::memory_order_relaxed);
          }
          t = 0;
        }
        if ((mine & 63) == 0) { steps.fetch_add(64, std::memory_order_relaxed); }
      }
    });
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& t : th) t.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double samples = (double)steps.load() / 16.0 + 1.0;
  printf("{\"agent_steps\": %lld, \"rollouts\": %lld, \"seconds\": %.3f, \"us_per_step\": {\"state_copy\": %.1f, \"submit\": %.1f, \"wait\": %.1f}}\n",
         steps.load(), rollouts.load(), dt, ns_copy.load() / samples * 1e-3, ns_submit.load() / samples * 1e-3, ns_wait.load() / samples * 1e-3);
  ga3c_shm_close(shm, 0);
  return 0;
}
