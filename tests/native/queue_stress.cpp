// Multi-threaded stress of the shared-memory transport (include/ga3c_host.h), built with
// -fsanitize=thread by tests/test_host_tsan.py.  Agents and predictors run as threads of ONE process on
// one segment: the protocol is the same as across processes, and TSAN sees every access.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/ga3c_host.h"

static std::atomic<int> failures{0};
static ga3c_shm* g_shm = nullptr;

// stands in for ga3c_net_predict_gather in the native predictor loop (ga3c_pq_serve)
static int fake_predict_rows(void*, const int64_t* offsets, int32_t batch, int32_t, float* p, float* v, float*) {
  const unsigned char* base = (const unsigned char*)ga3c_shm_base(g_shm);
  for (int i = 0; i < batch; ++i) {
    int tag;
    std::memcpy(&tag, base + offsets[i], 4);
    v[i] = (float)tag;
    for (int o = 0; o < 6; ++o) p[i * 6 + o] = (float)(tag % 7 + o);
  }
  return 0;
}
// ... and for ga3c_net_predict_gather_begin / _end in the pipelined loop (ga3c_pq_serve_pipelined): begin keeps the batch's
// offsets under a ticket (per calling thread: a thread has one batch begun at a time), end computes the answers
static thread_local int64_t t_offs[8];
static int fake_begin(void*, const int64_t* offsets, int32_t batch, int32_t, int32_t* ticket) {
  for (int i = 0; i < batch; ++i) t_offs[i] = offsets[i];
  *ticket = batch;
  return 0;
}
static int fake_end(void*, int32_t ticket, int32_t batch, float* p, float* v) {
  if (ticket != batch) return -1;
  return fake_predict_rows(nullptr, t_offs, batch, 1, p, v, nullptr);
}
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s line %d\n", #c, __LINE__); failures++; } } while (0)

int main(int argc, char** argv) {
  // queue_stress [rounds] [agents] [plain predictor threads]: the defaults oversubscribe lightly; "300 96 4" is a herd of
  // producers behind few consumers (the shape that collapsed the CAS-loop ring: every respond() wakes a batch of agents that
  // submit at once)
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 1500, n_agents = argc > 2 ? std::atoi(argv[2]) : 12,
            n_pred = argc > 3 ? std::atoi(argv[3]) : 3, n_act = 6;
  ga3c_shm_config cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.max_agents = n_agents; cfg.num_actions = n_act; cfg.state_bytes = 64; cfg.train_slots = 4; cfg.train_rows = 6;
  ga3c_shm* shm = nullptr;
  char name[64];
  std::snprintf(name, sizeof name, "/ga3c_tsan_%d", (int)getpid());
  REQUIRE(ga3c_shm_create(name, &cfg, &shm) == 0);
  g_shm = shm;
  {   // frame front-end under the sanitizers: odd geometries walk every bound of the two resample passes
    const int geo[5][3] = {{210, 160, 3}, {250, 160, 4}, {40, 50, 3}, {84, 84, 3}, {7, 300, 3}};
    for (auto& g : geo) {
      std::vector<uint8_t> rgb((size_t)g[0] * g[1] * g[2]), plane(84 * 84);
      for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (uint8_t)(i * 2654435761u >> 13);
      REQUIRE(ga3c_frame_preprocess(rgb.data(), g[0], g[1], g[2], 84, 84, plane.data()) == 0);
    }
  }
  std::atomic<long> served{0}, trained_rows{0}, produced_rows{0};
  std::vector<std::thread> th;
  th.emplace_back([&] {     // one predictor runs the native loop
    ga3c_serve_stats st;
    std::memset(&st, 0, sizeof st);
    int rc;
    while ((rc = ga3c_pq_serve(shm, fake_predict_rows, nullptr, 1, 8, 20, &st)) == GA3C_H_OK) {}
    REQUIRE(rc == GA3C_H_ECLOSED);
    REQUIRE(st.largest_batch <= 8);
    served += st.served;
  });
  th.emplace_back([&] {     // ... and one the pipelined native loop (answers batch k after batch k+1 has been begun)
    ga3c_serve_stats st;
    std::memset(&st, 0, sizeof st);
    int rc;
    while ((rc = ga3c_pq_serve_pipelined(shm, fake_begin, fake_end, nullptr, 1, 8, 20, &st)) == GA3C_H_OK) {}
    REQUIRE(rc == GA3C_H_ECLOSED);
    REQUIRE(st.largest_batch <= 8);
    served += st.served;
  });
  for (int p = 0; p < n_pred; ++p)
    th.emplace_back([&] {
      uint32_t ids[8];
      float pbuf[8 * 6], vbuf[8];
      for (;;) {
        const int n = ga3c_pq_pop_batch(shm, ids, 8, 50);
        if (n == GA3C_H_ECLOSED) return;
        if (n <= 0) continue;
        for (int i = 0; i < n; ++i) {
          const unsigned char* st = (const unsigned char*)ga3c_pq_state_ptr(shm, (int)ids[i]);
          int tag;
          std::memcpy(&tag, st, 4);
          vbuf[i] = (float)tag;
          for (int o = 0; o < n_act; ++o) pbuf[i * n_act + o] = (float)(tag % 7 + o);
        }
        REQUIRE(ga3c_pq_respond(shm, ids, n, pbuf, vbuf) == 0);
        served += n;
      }
    });
  std::thread trainer([&] {
    for (;;) {
      const int slot = ga3c_tq_pop(shm, 50);
      if (slot == GA3C_H_ECLOSED) return;
      if (slot < 0) continue;
      const int rows = ga3c_tq_rows(shm, slot);
      const float* ret = ga3c_tq_returns(shm, slot);
      const int32_t* act = ga3c_tq_actions(shm, slot);
      for (int i = 0; i < rows; ++i) REQUIRE((int)ret[i] == act[i] * 3);
      trained_rows += rows;
      REQUIRE(ga3c_tq_release(shm, slot) == 0);
    }
  });
  std::vector<std::thread> agents;
  for (int a = 0; a < n_agents; ++a)
    agents.emplace_back([&, a] {
      float p[6], v;
      for (int k = 0; k < rounds; ++k) {
        const int tag = a * 100000 + k;
        int rc;
        if ((a + k) % 3 == 0) {                 // the three calls of ProcessAgent.predict one by one ...
          std::memcpy(ga3c_pq_state_ptr(shm, a), &tag, 4);
          REQUIRE(ga3c_pq_submit(shm, a) == 0);
          while ((rc = ga3c_pq_wait(shm, a, p, &v, 100)) == GA3C_H_ETIMEOUT) {}
        } else {                                // ... or the whole step in one call (ga3c_pq_round_trip), with and without a draw
          int32_t action = -7;
          const double u = (a + k) % 3 == 1 ? 0.999999 : -1.0;
          rc = ga3c_pq_round_trip(shm, a, &tag, 4, 0, 1, 100, u, p, &v, &action);
          while (rc == GA3C_H_ETIMEOUT) rc = ga3c_pq_round_trip(shm, a, nullptr, 0, 0, 0, 100, u, p, &v, &action);
          if (rc == 0) REQUIRE(u < 0 ? action == -1 : (action >= 0 && action < 6));
        }
        REQUIRE(rc == 0);
        REQUIRE((int)v == tag);
        REQUIRE((int)p[2] == tag % 7 + 2);
        if (k % 5 == 4) {                       // ship a rollout
          int slot;
          while ((slot = ga3c_tq_acquire(shm, 100)) == GA3C_H_ETIMEOUT) {}
          REQUIRE(slot >= 0);
          const int rows = 1 + (k % 6);
          for (int i = 0; i < rows; ++i) { ga3c_tq_actions(shm, slot)[i] = a + i; ga3c_tq_returns(shm, slot)[i] = (float)((a + i) * 3); }
          REQUIRE(ga3c_tq_commit(shm, slot, rows) == 0);
          produced_rows += rows;
        }
      }
    });
  if (n_agents > 1) ga3c_pq_set_spin(shm, 20);     // agents 0.. poll briefly before they announce their sleep (the waiter flag's other path)
  for (auto& t : agents) t.join();
  while (trained_rows.load() < produced_rows.load()) std::this_thread::yield();
  ga3c_shm_shutdown(shm);
  for (auto& t : th) t.join();
  trainer.join();
  REQUIRE(served.load() == (long)n_agents * rounds);
  REQUIRE(trained_rows.load() == produced_rows.load());
  ga3c_shm_close(shm, 1);
  std::printf("served %ld predictions, %ld rollout rows, failures %d\n", served.load(), trained_rows.load(), failures.load());
  return failures.load() ? 1 : 0;
}
