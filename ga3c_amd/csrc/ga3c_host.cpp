// ga3c_host.cpp -- host-only half of the GA3C hot path: bit-exact returns and the shared-memory
// transport that replaces the reference's pickling multiprocessing.Queues (see include/ga3c_host.h).
// No HIP here: agent processes load only this library.
#include "../../include/ga3c_host.h"

#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "ga3c_resample.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[400];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// ---- futex helpers (cross-process: no FUTEX_PRIVATE_FLAG)
int futex_wait(std::atomic<uint32_t>* addr, uint32_t expect, int timeout_ms) {
  struct timespec ts, *pts = nullptr;
  if (timeout_ms >= 0) {
    ts.tv_sec = timeout_ms / 1000;
    ts.tv_nsec = (long)(timeout_ms % 1000) * 1000000L;
    pts = &ts;
  }
  return (int)syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAIT, expect, pts, nullptr, 0);
}
void futex_wake(std::atomic<uint32_t>* addr, int n) {
  syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAKE, n, nullptr, nullptr, 0);
}
int64_t now_ns() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}
// the batching threads carry names (`top -H`, /proc/<pid>/task/*/comm): where the server's CPU time goes is read per name
void name_this_thread(const char* name) {
  thread_local const char* current = nullptr;
  if (current == name) return;
  current = name;
  prctl(PR_SET_NAME, (unsigned long)name, 0, 0, 0);
}
int64_t now_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}

// ---- bounded lock-free MPMC ring of u32 (sequence-per-cell scheme), laid out inside the segment
struct Cell {
  std::atomic<uint64_t> seq;
  uint32_t data;
  uint32_t pad;
};
struct Ring {
  alignas(64) std::atomic<uint64_t> enq;
  alignas(64) std::atomic<uint64_t> deq;
  alignas(64) std::atomic<uint32_t> signal;   // bumped on every push; the futex word consumers sleep on
  std::atomic<uint32_t> waiters;
  uint32_t mask;
  std::atomic<uint32_t> wake_claim;  // 1 while a wake-up of a sleeping consumer is under way: further pushes make no system call
  int64_t cells_off;   // byte offset of the cell array from the segment base
};

Cell* cells(char* base, Ring* r) { return reinterpret_cast<Cell*>(base + r->cells_off); }

void ring_init(char* base, Ring* r, uint32_t cap_pow2, int64_t cells_off) {
  r->enq.store(0);
  r->deq.store(0);
  r->signal.store(0);
  r->waiters.store(0);
  r->wake_claim.store(0);
  r->mask = cap_pow2 - 1;
  r->cells_off = cells_off;
  Cell* c = cells(base, r);
  for (uint32_t i = 0; i < cap_pow2; ++i) {
    c[i].seq.store(i);
    c[i].data = 0;
  }
}

// Push = take a ticket (ONE fetch_add on enq, never retried), then fill the ticket's cell.  A compare-and-swap loop on
// enq -- the textbook form of this ring -- collapses under a herd of producers: a predictor's respond() wakes a hundred
// agents at once, they all submit within microseconds, and each failed CAS is another round of the contended cache line
// (measured with 256 agent threads on a 2-socket host: 41 attempts and 350 us per push, 15 cores burnt spinning, 50 k
// requests/s instead of 500 k).  The rings are sized so that they are never logically full (one request per agent, one
// entry per rollout slot, capacity many times that), so a ticket's cell is free unless the consumer of the previous lap
// has claimed it (CAS on deq) and was descheduled before republishing its sequence number: that is a wait for a peer
// thread -- yield a few times, then nap with a doubling back-off; `closed` ends it.
//
// A ticket that has been taken must be filled: a producer that dies between the fetch_add and the sequence store leaves
// a hole no consumer can pass.  Server.remove_agent ends an agent that does not leave by itself with SIGTERM
// (Process.terminate), so asynchronous signals are held back for the few instructions between ticket and publication
// (two rt_sigprocmask calls, ~0.2 us, against an agent step of ~28 us); SIGKILL cannot be held back and may still wedge
// a ring -- the transport has to be recreated then.  On `closed` the cell is abandoned on purpose: every consumer treats
// a closed segment as drained.
// (ga3c_host_signal_hold(0) switches the mask off for a process whose producers are THREADS: rt_sigprocmask takes the
// process's sighand lock, which its threads share -- 256 to 512 agent threads submitting a million times a second turned it
// into 10-13 cores of system time; agent PROCESSES, the product's, have a lock each and keep the default.)
std::atomic<int> g_signal_hold{1};

struct SignalHold {
  sigset_t old;
  bool on;
  SignalHold() : on(g_signal_hold.load(std::memory_order_relaxed) != 0) {
    if (!on) return;
    sigset_t all;
    sigfillset(&all);
    pthread_sigmask(SIG_BLOCK, &all, &old);
  }
  ~SignalHold() { if (on) pthread_sigmask(SIG_SETMASK, &old, nullptr); }
};

bool ring_push(char* base, Ring* r, uint32_t v, const std::atomic<uint32_t>* closed) {
  Cell* c = cells(base, r);
  SignalHold hold;
  const uint64_t pos = r->enq.fetch_add(1, std::memory_order_relaxed);
  Cell* cell = &c[pos & r->mask];
  int64_t nap_ns = 50000;
  for (int spin = 0; cell->seq.load(std::memory_order_acquire) != pos; ++spin) {
    if (spin < 8) {
      sched_yield();
    } else {
      if (closed && closed->load(std::memory_order_acquire)) return false;   // shutting down: nobody will pop anyway
      struct timespec ts = {0, (long)nap_ns};
      nanosleep(&ts, nullptr);
      if (nap_ns < 2000000) nap_ns *= 2;
    }
  }
  cell->data = v;
  cell->seq.store(pos + 1, std::memory_order_release);
  r->signal.fetch_add(1, std::memory_order_seq_cst);
  // One wake-up per sleeper, not one per push: a predictor's answers wake a hundred agents within microseconds, they all
  // submit while the consumer they find asleep is still on its way out of FUTEX_WAIT (~4 us), and a FUTEX_WAKE from each of
  // them is a hundred threads on one futex hash bucket's lock (measured: 7-10 cores of system time in the agents, `submit`
  // 4-9 us instead of 0.8, the cgroup throttled).  The first push claims the wake-up; a consumer drops the claim when it is
  // back from its wait and again before it goes to sleep, so a claim that found nobody asleep cannot outlive the next sleeper.
  if (r->waiters.load(std::memory_order_seq_cst) != 0 && r->wake_claim.load(std::memory_order_seq_cst) == 0 &&
      r->wake_claim.exchange(1, std::memory_order_seq_cst) == 0)
    futex_wake(&r->signal, 1);
  return true;
}

bool ring_try_pop(char* base, Ring* r, uint32_t* v) {
  Cell* c = cells(base, r);
  uint64_t pos = r->deq.load(std::memory_order_relaxed);
  for (;;) {
    Cell* cell = &c[pos & r->mask];
    const uint64_t seq = cell->seq.load(std::memory_order_acquire);
    const int64_t dif = (int64_t)seq - (int64_t)(pos + 1);
    if (dif == 0) {
      if (r->deq.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
        *v = cell->data;
        cell->seq.store(pos + r->mask + 1, std::memory_order_release);
        return true;
      }
    } else if (dif < 0) {
      return false;   // empty
    } else {
      pos = r->deq.load(std::memory_order_relaxed);
    }
  }
}

uint32_t ring_size(Ring* r) {
  const uint64_t e = r->enq.load(std::memory_order_acquire), d = r->deq.load(std::memory_order_acquire);
  return e > d ? (uint32_t)(e - d) : 0;
}

constexpr uint64_t MAGIC = 0x4741334353484d31ull;   // "GA3CSHM1"
constexpr int MAXA = 64;

struct AgentMeta {   // lives right behind each agent's state bytes
  float p[MAXA];
  float v;
  uint32_t req_seq;                  // written by the agent only
  std::atomic<uint32_t> resp_seq;    // futex word: predictor -> agent
  uint32_t req_flags;                // written by the agent before it submits (GA3C_REQ_*), read by the predictor
  uint32_t req_epoch;                // times req_seq has wrapped (written by the agent only): request number = epoch << 32 | seq
  std::atomic<uint32_t> waiting;     // 1 while the agent is (about to be) asleep on resp_seq: only then does an answer cost a syscall
  uint32_t pad0;
  // answer -> agent running again (ga3c_pq_wake_latency): stamped by the predictor, summed by the agent; CLOCK_MONOTONIC, ns
  uint64_t answered_ns;              // when ga3c_pq_respond published the newest answer
  uint64_t lat_sum_ns[2];            // [0] the answer was there or came while polling, [1] the agent had gone to sleep on the futex
  uint32_t lat_count[2];
  uint32_t lat_max_ns[2];
  uint32_t pad[46];
};
static_assert(sizeof(AgentMeta) == 512, "AgentMeta must stay 512 bytes");

struct Header {
  uint64_t magic;
  ga3c_shm_config cfg;
  int64_t total_bytes;
  int64_t agents_off, agent_stride, state_span;
  int64_t rollouts_off, rollout_stride, ro_returns_off, ro_actions_off, ro_rows_off;
  std::atomic<uint32_t> closed;
  uint32_t pad;
  std::atomic<int32_t> linger_us;      // ga3c_pq_set_linger: how long a predictor keeps collecting after the first request
  std::atomic<int32_t> linger_batch;   // ... unless it already holds this many
  std::atomic<int32_t> spin_us;        // ga3c_pq_set_spin: how long an agent polls for its answer before it sleeps
  int32_t pad2;
  Ring req, freeq, readyq;
};

int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
uint32_t pow2_at_least(uint32_t v) {
  uint32_t p = 2;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

struct ga3c_shm {
  char* base = nullptr;
  int64_t bytes = 0;
  std::string name;
  bool owner = false;
  Header* hdr() const { return reinterpret_cast<Header*>(base); }
  AgentMeta* meta(int agent) const {
    return reinterpret_cast<AgentMeta*>(base + hdr()->agents_off + agent * hdr()->agent_stride + hdr()->state_span);
  }
  char* rollout(int slot) const { return base + hdr()->rollouts_off + slot * hdr()->rollout_stride; }
};

namespace {

// Block until an item can be popped from `r`, the segment is closed, or the timeout passes.
// pass_on: a consumer that slept and then got an entry wakes the next sleeper if more entries are there -- for rings whose
// consumers take ONE entry each (the free rollout slots: a trainer gives back two dozen at a time and its pushes wake one
// agent, see ring_push); the rings drained by a batching thread leave the others asleep.
int ring_pop_wait(ga3c_shm* s, Ring* r, uint32_t* v, int timeout_ms, bool pass_on = false) {
  Header* h = s->hdr();
  const int64_t deadline = timeout_ms >= 0 ? now_ms() + timeout_ms : -1;
  bool slept = false;
  for (;;) {
    if (ring_try_pop(s->base, r, v)) {
      if (pass_on && slept && r->waiters.load(std::memory_order_seq_cst) != 0 && ring_size(r) != 0) futex_wake(&r->signal, 1);
      return GA3C_H_OK;
    }
    if (h->closed.load(std::memory_order_acquire)) return GA3C_H_ECLOSED;
    const uint32_t sig = r->signal.load(std::memory_order_seq_cst);
    r->waiters.fetch_add(1, std::memory_order_seq_cst);
    r->wake_claim.store(0, std::memory_order_seq_cst);      // whoever pushes from here on may wake this consumer
    int rc = GA3C_H_OK;
    if (ring_try_pop(s->base, r, v)) {
      r->waiters.fetch_sub(1, std::memory_order_seq_cst);
      return GA3C_H_OK;
    }
    int wait_ms = -1;
    if (deadline >= 0) {
      const int64_t left = deadline - now_ms();
      if (left <= 0) rc = GA3C_H_ETIMEOUT;
      wait_ms = (int)left;
    }
    if (rc == GA3C_H_OK) { futex_wait(&r->signal, sig, wait_ms); slept = true; }
    r->wake_claim.store(0, std::memory_order_seq_cst);
    r->waiters.fetch_sub(1, std::memory_order_seq_cst);
    if (rc != GA3C_H_OK) return rc;
  }
}

}  // namespace

extern "C" {

const char* ga3c_host_last_error(void) { return g_err.c_str(); }

int ga3c_host_signal_hold(int32_t on) {
  g_signal_hold.store(on ? 1 : 0, std::memory_order_relaxed);
  return GA3C_H_OK;
}

int ga3c_returns_fork(const double* rewards, int32_t T, double gamma, double terminal_reward, int32_t discounting,
                      int32_t use_intermediate_reward, double* out) {
  if (T < 0 || (T > 0 && (!rewards || !out))) return fail(GA3C_H_EINVAL, "bad arguments");
  for (int32_t t = 0; t < T; ++t) out[t] = rewards[t];
  double reward_sum = terminal_reward;
  for (int32_t t = T - 2; t >= 0; --t) {
    if (discounting) {
      reward_sum = gamma * reward_sum;
      if (!use_intermediate_reward) out[t] = reward_sum;
      // with intermediate rewards the reference folds r into reward_sum and stores nothing
    }
  }
  return GA3C_H_OK;
}

int ga3c_returns_nstep(const double* rewards, int32_t T, double gamma, double bootstrap_value, double rmin,
                       double rmax, double* out) {
  if (T < 0 || (T > 1 && (!rewards || !out))) return fail(GA3C_H_EINVAL, "bad arguments");
  double reward_sum = bootstrap_value;
  for (int32_t t = T - 2; t >= 0; --t) {
    double r = rewards[t];
    r = r < rmin ? rmin : (r > rmax ? rmax : r);
    reward_sum = gamma * reward_sum + r;
    out[t] = reward_sum;
  }
  return GA3C_H_OK;
}

// ---- host frame front-end.  Same arithmetic as oracle/frame_frontend.py and frame_frontend_kernel, arranged for speed:
// resample tables cached per geometry, scratch buffers kept per thread, and the three hot loops compiled twice -- for
// x86-64-v3 (AVX2 + FMA: the compiler vectorises them, std::fma is one instruction) and for the baseline ISA -- with the
// choice made once at run time.  Results do not depend on the path: every product and sum is rounded where numpy
// rounds it (-ffp-contract=off, explicit fma only in the gray product).
namespace {

struct FrameScratch {
  std::vector<double> gray;
  std::vector<uint8_t> img, tmp;
  ga3c::ResampleTable th, tv;
  // horizontal pass, 4 outputs per step: first source byte of the group, and per tap a byte-shuffle mask that drops each
  // output's tap byte into its 32-bit lane plus the four weights; empty when a group's sources span more than 16 bytes
  std::vector<int32_t> hbase, hweight;   // [groups], [groups][ksize][4]
  std::vector<uint8_t> hsel;             // [groups][ksize][16]
};

void build_hpass_plan(FrameScratch& sc, int out_w) {
  const ga3c::ResampleTable& t = sc.th;
  sc.hbase.clear(); sc.hweight.clear(); sc.hsel.clear();
  if (out_w % 4) return;
  const int groups = out_w / 4;
  std::vector<int32_t> base((size_t)groups), weight((size_t)groups * t.ksize * 4, 0);
  std::vector<uint8_t> sel((size_t)groups * t.ksize * 16, 0x80);
  for (int g = 0; g < groups; ++g) {
    base[g] = t.bounds[(size_t)(4 * g) * 2];
    for (int q = 0; q < 4; ++q) {
      const int xx = 4 * g + q, xmin = t.bounds[(size_t)xx * 2], n = t.bounds[(size_t)xx * 2 + 1];
      if (xmin < base[g] || xmin + n - base[g] > 16) return;          // does not fit one 16-byte window: scalar pass
      for (int k = 0; k < n; ++k) {
        sel[((size_t)g * t.ksize + k) * 16 + 4 * q] = (uint8_t)(xmin - base[g] + k);
        weight[((size_t)g * t.ksize + k) * 4 + q] = t.kk[(size_t)xx * t.ksize + k];
      }
    }
  }
  sc.hbase.swap(base); sc.hweight.swap(weight); sc.hsel.swap(sel);
}

#define GA3C_FE_LOOPS(SUFFIX, ATTR)                                                                                    \
  ATTR void fe_gray##SUFFIX(const uint8_t* rgb, size_t npx, int channels, double* gray, double* lo, double* hi) {      \
    double mn = 1e300, mx = -1e300;                                                                                    \
    size_t i = 0;                                                                                                      \
    if (channels == 3) {        /* 4 pixels = 12 contiguous bytes per turn: independent chains, min / max folded late */ \
      for (; i + 4 <= npx; i += 4) {                                                                                   \
        const uint8_t* px = rgb + i * 3;                                                                               \
        const double g0 = std::fma((double)px[2], 0.114, std::fma((double)px[1], 0.587, (double)px[0] * 0.299));       \
        const double g1 = std::fma((double)px[5], 0.114, std::fma((double)px[4], 0.587, (double)px[3] * 0.299));       \
        const double g2 = std::fma((double)px[8], 0.114, std::fma((double)px[7], 0.587, (double)px[6] * 0.299));       \
        const double g3 = std::fma((double)px[11], 0.114, std::fma((double)px[10], 0.587, (double)px[9] * 0.299));     \
        gray[i] = g0; gray[i + 1] = g1; gray[i + 2] = g2; gray[i + 3] = g3;                                            \
        const double a = g0 < g1 ? g0 : g1, b = g2 < g3 ? g2 : g3, c = g0 > g1 ? g0 : g1, d = g2 > g3 ? g2 : g3;       \
        const double lo4 = a < b ? a : b, hi4 = c > d ? c : d;                                                         \
        mn = lo4 < mn ? lo4 : mn;                                                                                      \
        mx = hi4 > mx ? hi4 : mx;                                                                                      \
      }                                                                                                                \
    }                                                                                                                  \
    for (; i < npx; ++i) {                                                                                             \
      const uint8_t* px = rgb + i * channels;                                                                          \
      /* np.dot's order on a [H, W, 3] frame: fused multiply-adds, left to right */                                    \
      const double g = std::fma((double)px[2], 0.114, std::fma((double)px[1], 0.587, (double)px[0] * 0.299));          \
      gray[i] = g;                                                                                                     \
      mn = g < mn ? g : mn;                                                                                            \
      mx = g > mx ? g : mx;                                                                                            \
    }                                                                                                                  \
    *lo = mn;                                                                                                          \
    *hi = mx;                                                                                                          \
  }                                                                                                                    \
  ATTR void fe_bytescale##SUFFIX(const double* gray, size_t npx, double cmin, double scale, uint8_t* img) {            \
    for (size_t i = 0; i < npx; ++i) {                                                                                 \
      const double d = gray[i] - cmin;         /* separately rounded subtract and multiply, as numpy evaluates them */ \
      const double t = d * scale;                                                                                      \
      const double c = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t);                                                        \
      img[i] = (uint8_t)(int)(c + 0.5);                                                                                \
    }                                                                                                                  \
  }                                                                                                                    \
  ATTR void fe_vpass##SUFFIX(const uint8_t* src, int cur_w, const ga3c::ResampleTable& t, uint8_t* plane) {            \
    std::vector<int32_t> acc((size_t)cur_w);                                                                           \
    for (int yy = 0; yy < t.out_size; ++yy) {                                                                          \
      const int ymin = t.bounds[(size_t)yy * 2], n = t.bounds[(size_t)yy * 2 + 1];                                     \
      for (int x = 0; x < cur_w; ++x) acc[x] = 1 << (ga3c::RESAMPLE_PRECISION_BITS - 1);                               \
      for (int k = 0; k < n; ++k) {                                                                                    \
        const uint8_t* row = src + (size_t)(ymin + k) * cur_w;                                                         \
        const int32_t c = t.kk[(size_t)yy * t.ksize + k];                                                              \
        for (int x = 0; x < cur_w; ++x) acc[x] += (int32_t)row[x] * c;                                                 \
      }                                                                                                                \
      for (int x = 0; x < cur_w; ++x) {                                                                                \
        const int32_t v = acc[x] >> ga3c::RESAMPLE_PRECISION_BITS;                                                     \
        plane[(size_t)yy * cur_w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));                                    \
      }                                                                                                                \
    }                                                                                                                  \
  }

GA3C_FE_LOOPS(_base, )
#if defined(__x86_64__)
GA3C_FE_LOOPS(_auto, __attribute__((target("avx2,fma"))))

// hand-vectorised forms of the two per-pixel loops (4 pixels per step): same operations in the same order per pixel
__attribute__((target("avx2,fma"))) void fe_gray_v3(const uint8_t* rgb, size_t npx, int channels, double* gray, double* lo,
                                                    double* hi) {
  size_t i = 0;
  __m256d vmn = _mm256_set1_pd(1e300), vmx = _mm256_set1_pd(-1e300);
  if (channels == 3 && npx >= 8) {
    const __m128i sr = _mm_setr_epi8(0, -1, -1, -1, 3, -1, -1, -1, 6, -1, -1, -1, 9, -1, -1, -1);
    const __m128i sg = _mm_setr_epi8(1, -1, -1, -1, 4, -1, -1, -1, 7, -1, -1, -1, 10, -1, -1, -1);
    const __m128i sb = _mm_setr_epi8(2, -1, -1, -1, 5, -1, -1, -1, 8, -1, -1, -1, 11, -1, -1, -1);
    const __m256d c0 = _mm256_set1_pd(0.299), c1 = _mm256_set1_pd(0.587), c2 = _mm256_set1_pd(0.114);
    for (; i + 8 <= npx; i += 4) {     // the 16-byte load covers 12 pixel bytes + 4 more: stop two groups early
      const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(rgb + i * 3));
      const __m256d r = _mm256_cvtepi32_pd(_mm_shuffle_epi8(v, sr));
      const __m256d g = _mm256_cvtepi32_pd(_mm_shuffle_epi8(v, sg));
      const __m256d b = _mm256_cvtepi32_pd(_mm_shuffle_epi8(v, sb));
      const __m256d y = _mm256_fmadd_pd(b, c2, _mm256_fmadd_pd(g, c1, _mm256_mul_pd(r, c0)));
      _mm256_storeu_pd(gray + i, y);
      vmn = _mm256_min_pd(vmn, y);
      vmx = _mm256_max_pd(vmx, y);
    }
  }
  double t[4], mn, mx;
  _mm256_storeu_pd(t, vmn);
  mn = std::min(std::min(t[0], t[1]), std::min(t[2], t[3]));
  _mm256_storeu_pd(t, vmx);
  mx = std::max(std::max(t[0], t[1]), std::max(t[2], t[3]));
  for (; i < npx; ++i) {
    const uint8_t* px = rgb + i * channels;
    const double g = std::fma((double)px[2], 0.114, std::fma((double)px[1], 0.587, (double)px[0] * 0.299));
    gray[i] = g;
    mn = g < mn ? g : mn;
    mx = g > mx ? g : mx;
  }
  *lo = mn;
  *hi = mx;
}

__attribute__((target("avx2,fma"))) void fe_bytescale_v3(const double* gray, size_t npx, double cmin, double scale, uint8_t* img) {
  size_t i = 0;
  const __m256d vmin = _mm256_set1_pd(cmin), vs = _mm256_set1_pd(scale), z = _mm256_setzero_pd(), top = _mm256_set1_pd(255.0),
                half = _mm256_set1_pd(0.5);
  for (; i + 4 <= npx; i += 4) {
    __m256d t = _mm256_mul_pd(_mm256_sub_pd(_mm256_loadu_pd(gray + i), vmin), vs);   // two roundings, as numpy
    t = _mm256_min_pd(_mm256_max_pd(t, z), top);
    const __m128i q = _mm256_cvttpd_epi32(_mm256_add_pd(t, half));                    // truncation, values in 0..255
    const __m128i b = _mm_packus_epi16(_mm_packus_epi32(q, q), q);
    const int w = _mm_cvtsi128_si32(b);
    memcpy(img + i, &w, 4);
  }
  for (; i < npx; ++i) {
    const double d = gray[i] - cmin;
    const double t = d * scale;
    const double c = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t);
    img[i] = (uint8_t)(int)(c + 0.5);
  }
}

// horizontal pass with the plan of build_hpass_plan: per 4 outputs one 16-byte load, then per tap a shuffle, a 32-bit
// multiply and an add -- the integer arithmetic of the scalar loop, lane by lane
__attribute__((target("avx2,fma"))) void fe_hpass_v3(const uint8_t* src, int height, int width, int out_w, const FrameScratch& sc,
                                                     uint8_t* dst) {
  const int groups = out_w / 4, ks = sc.th.ksize;
  const __m128i half = _mm_set1_epi32(1 << (ga3c::RESAMPLE_PRECISION_BITS - 1));
  for (int y = 0; y < height; ++y) {
    const uint8_t* row = src + (size_t)y * width;
    uint8_t* out = dst + (size_t)y * out_w;
    for (int g = 0; g < groups; ++g) {
      const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + sc.hbase[g]));   // buffers carry 16 spare bytes
      __m128i acc = half;
      for (int k = 0; k < ks; ++k) {
        const __m128i sel = _mm_loadu_si128(reinterpret_cast<const __m128i*>(&sc.hsel[((size_t)g * ks + k) * 16]));
        const __m128i w = _mm_loadu_si128(reinterpret_cast<const __m128i*>(&sc.hweight[((size_t)g * ks + k) * 4]));
        acc = _mm_add_epi32(acc, _mm_mullo_epi32(_mm_shuffle_epi8(v, sel), w));
      }
      acc = _mm_srai_epi32(acc, ga3c::RESAMPLE_PRECISION_BITS);
      const __m128i b = _mm_packus_epi16(_mm_packus_epi32(acc, acc), acc);     // saturating: clip8
      const int w4 = _mm_cvtsi128_si32(b);
      memcpy(out + 4 * g, &w4, 4);
    }
  }
}
#endif
#undef GA3C_FE_LOOPS

bool cpu_has_v3() {
#if defined(__x86_64__)
  static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  return ok;
#else
  return false;
#endif
}

}  // namespace

int ga3c_frame_preprocess(const uint8_t* rgb, int32_t height, int32_t width, int32_t channels, int32_t out_h,
                          int32_t out_w, uint8_t* plane) {
  if (!rgb || !plane || height < 1 || width < 1 || channels < 3 || out_h < 1 || out_w < 1)
    return fail(GA3C_H_EINVAL, "bad argument");
  thread_local FrameScratch sc;
  const size_t npx = (size_t)height * width;
  if (sc.gray.size() < npx) { sc.gray.resize(npx); sc.img.assign(npx + 16, 0); }   // + 16: the vector pass loads whole windows
  if (sc.th.in_size != width || sc.th.out_size != out_w) {
    sc.th = ga3c::make_bilinear_table(width, out_w);
    build_hpass_plan(sc, out_w);
  }
  if (sc.tv.in_size != height || sc.tv.out_size != out_h) sc.tv = ga3c::make_bilinear_table(height, out_h);
  const bool v3 = cpu_has_v3();
  double cmin, cmax;
#if defined(__x86_64__)
  if (v3) fe_gray_v3(rgb, npx, channels, sc.gray.data(), &cmin, &cmax); else
#endif
    fe_gray_base(rgb, npx, channels, sc.gray.data(), &cmin, &cmax);
  double cscale = cmax - cmin;
  if (cscale == 0.0) cscale = 1.0;
  const double scale = 255.0 / cscale;
#if defined(__x86_64__)
  if (v3) fe_bytescale_v3(sc.gray.data(), npx, cmin, scale, sc.img.data()); else
#endif
    fe_bytescale_base(sc.gray.data(), npx, cmin, scale, sc.img.data());
  const uint8_t* src = sc.img.data();
  int cur_w = width;
  if (width != out_w) {   // horizontal pass: few taps per output, not worth vectorising
    const ga3c::ResampleTable& t = sc.th;
    if (sc.tmp.size() < (size_t)height * out_w) sc.tmp.resize((size_t)height * out_w);
#if defined(__x86_64__)
    if (v3 && !sc.hbase.empty()) fe_hpass_v3(src, height, width, out_w, sc, sc.tmp.data()); else
#endif
    for (int y = 0; y < height; ++y) {
      const uint8_t* row = src + (size_t)y * width;
      uint8_t* dst = sc.tmp.data() + (size_t)y * out_w;
      for (int xx = 0; xx < out_w; ++xx) {
        const int xmin = t.bounds[(size_t)xx * 2], n = t.bounds[(size_t)xx * 2 + 1];
        const int32_t* kk = &t.kk[(size_t)xx * t.ksize];
        int32_t acc = 1 << (ga3c::RESAMPLE_PRECISION_BITS - 1);
        if (t.ksize == 5 && xmin + 5 <= width) {   // the Atari geometry (160 -> 84): taps past the count are 0
          acc += (int32_t)row[xmin] * kk[0] + (int32_t)row[xmin + 1] * kk[1] + (int32_t)row[xmin + 2] * kk[2] +
                 (int32_t)row[xmin + 3] * kk[3] + (int32_t)row[xmin + 4] * kk[4];
        } else {
          for (int x = 0; x < n; ++x) acc += (int32_t)row[xmin + x] * kk[x];
        }
        const int32_t v = acc >> ga3c::RESAMPLE_PRECISION_BITS;
        dst[xx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    }
    src = sc.tmp.data();
    cur_w = out_w;
  }
  if (height != out_h) {  // vertical pass: whole rows at a time
#if defined(__x86_64__)
    if (v3) fe_vpass_auto(src, cur_w, sc.tv, plane); else
#endif
      fe_vpass_base(src, cur_w, sc.tv, plane);
  } else {
    memcpy(plane, src, (size_t)out_h * cur_w);
  }
  return GA3C_H_OK;
}

int ga3c_shm_create(const char* name, const ga3c_shm_config* cfg, ga3c_shm** out) {
  if (!name || !cfg || !out) return fail(GA3C_H_EINVAL, "null argument");
  if (cfg->max_agents < 1 || cfg->max_agents > 65536 || cfg->num_actions < 1 || cfg->num_actions > MAXA ||
      cfg->state_bytes < 16 || cfg->state_bytes % 16 != 0 || cfg->train_slots < 1 || cfg->train_slots > 65536 ||
      cfg->train_rows < 1 || cfg->rollout_row_bytes < 0 || cfg->rollout_row_bytes % 16 != 0)
    return fail(GA3C_H_EINVAL, "bad shm config");
  const int64_t row_bytes = cfg->rollout_row_bytes ? cfg->rollout_row_bytes : cfg->state_bytes;
  // well above the logical maximum (16 B per cell): a cell whose consumer is momentarily descheduled blocks the producers
  // only when they come round to it again, i.e. after `capacity` further pushes -- at 500 k requests/s a ring of 2 x
  // max_agents = 512 cells came round in one millisecond
  const uint32_t req_cap = pow2_at_least(16u * (uint32_t)cfg->max_agents < 4096u ? 4096u : 16u * (uint32_t)cfg->max_agents);
  const uint32_t tr_cap = pow2_at_least(8u * (uint32_t)cfg->train_slots < 256u ? 256u : 8u * (uint32_t)cfg->train_slots);
  Header lay;
  memset((void*)&lay, 0, sizeof lay);
  int64_t off = round_up(sizeof(Header), 256);
  const int64_t req_cells = off;   off = round_up(off + (int64_t)req_cap * sizeof(Cell), 256);
  const int64_t free_cells = off;  off = round_up(off + (int64_t)tr_cap * sizeof(Cell), 256);
  const int64_t ready_cells = off; off = round_up(off + (int64_t)tr_cap * sizeof(Cell), 4096);
  lay.state_span = round_up(cfg->state_bytes, 256);
  lay.agent_stride = lay.state_span + (int64_t)sizeof(AgentMeta);
  lay.agents_off = off;
  off = round_up(off + lay.agent_stride * cfg->max_agents, 4096);
  lay.ro_returns_off = round_up((int64_t)cfg->train_rows * row_bytes, 256);
  lay.ro_actions_off = lay.ro_returns_off + round_up((int64_t)cfg->train_rows * 4, 64);
  lay.ro_rows_off = lay.ro_actions_off + round_up((int64_t)cfg->train_rows * 4, 64);
  lay.rollout_stride = round_up(lay.ro_rows_off + 64, 256);
  lay.rollouts_off = off;
  off = round_up(off + lay.rollout_stride * cfg->train_slots, 4096);
  lay.total_bytes = off;

  shm_unlink(name);
  const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return fail(GA3C_H_ESYS, "shm_open(%s): %s", name, strerror(errno));
  if (ftruncate(fd, lay.total_bytes) != 0) {
    const int e = errno;
    close(fd);
    shm_unlink(name);
    return fail(GA3C_H_ESYS, "ftruncate(%lld): %s", (long long)lay.total_bytes, strerror(e));
  }
  void* p = mmap(nullptr, (size_t)lay.total_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    const int e = errno;
    shm_unlink(name);
    return fail(GA3C_H_ESYS, "mmap: %s", strerror(e));
  }
  ga3c_shm* s = new (std::nothrow) ga3c_shm();
  if (!s) {
    munmap(p, (size_t)lay.total_bytes);
    shm_unlink(name);
    return fail(GA3C_H_EINVAL, "out of memory");
  }
  s->base = (char*)p;
  s->bytes = lay.total_bytes;
  s->name = name;
  s->owner = true;
  Header* h = s->hdr();   // fresh shm pages are zero
  h->cfg = *cfg;
  h->total_bytes = lay.total_bytes;
  h->agents_off = lay.agents_off; h->agent_stride = lay.agent_stride; h->state_span = lay.state_span;
  h->rollouts_off = lay.rollouts_off; h->rollout_stride = lay.rollout_stride;
  h->ro_returns_off = lay.ro_returns_off; h->ro_actions_off = lay.ro_actions_off; h->ro_rows_off = lay.ro_rows_off;
  h->closed.store(0);
  ring_init(s->base, &h->req, req_cap, req_cells);
  ring_init(s->base, &h->freeq, tr_cap, free_cells);
  ring_init(s->base, &h->readyq, tr_cap, ready_cells);
  for (int i = 0; i < cfg->train_slots; ++i) ring_push(s->base, &h->freeq, (uint32_t)i, nullptr);
  std::atomic_thread_fence(std::memory_order_seq_cst);
  h->magic = MAGIC;
  *out = s;
  return GA3C_H_OK;
}

int ga3c_shm_attach(const char* name, ga3c_shm** out) {
  if (!name || !out) return fail(GA3C_H_EINVAL, "null argument");
  const int fd = shm_open(name, O_RDWR, 0600);
  if (fd < 0) return fail(GA3C_H_ESYS, "shm_open(%s): %s", name, strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < (off_t)sizeof(Header)) {
    close(fd);
    return fail(GA3C_H_ESYS, "segment %s too small", name);
  }
  void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail(GA3C_H_ESYS, "mmap: %s", strerror(errno));
  Header* h = (Header*)p;
  if (h->magic != MAGIC || h->total_bytes != (int64_t)st.st_size) {
    munmap(p, (size_t)st.st_size);
    return fail(GA3C_H_EINVAL, "segment %s is not a ga3c transport", name);
  }
  ga3c_shm* s = new (std::nothrow) ga3c_shm();
  if (!s) {
    munmap(p, (size_t)st.st_size);
    return fail(GA3C_H_EINVAL, "out of memory");
  }
  s->base = (char*)p;
  s->bytes = st.st_size;
  s->name = name;
  *out = s;
  return GA3C_H_OK;
}

int ga3c_shm_close(ga3c_shm* shm, int32_t unlink_segment) {
  if (!shm) return GA3C_H_OK;
  munmap(shm->base, (size_t)shm->bytes);
  if (unlink_segment) shm_unlink(shm->name.c_str());
  delete shm;
  return GA3C_H_OK;
}

int ga3c_shm_unlink(ga3c_shm* shm) {
  if (!shm) return fail(GA3C_H_EINVAL, "null argument");
  if (shm->owner) shm_unlink(shm->name.c_str());      // the name goes; mappings (ours, the agents', the GPU's) stay valid
  return GA3C_H_OK;
}

int ga3c_shm_shutdown(ga3c_shm* shm) {
  if (!shm) return fail(GA3C_H_EINVAL, "null argument");
  Header* h = shm->hdr();
  h->closed.store(1, std::memory_order_seq_cst);
  for (Ring* r : {&h->req, &h->freeq, &h->readyq}) {
    r->signal.fetch_add(1, std::memory_order_seq_cst);
    futex_wake(&r->signal, INT32_MAX);
  }
  for (int a = 0; a < h->cfg.max_agents; ++a) futex_wake(&shm->meta(a)->resp_seq, INT32_MAX);
  return GA3C_H_OK;
}

void* ga3c_shm_base(ga3c_shm* shm) { return shm ? shm->base : nullptr; }
int64_t ga3c_shm_bytes(ga3c_shm* shm) { return shm ? shm->bytes : 0; }

int ga3c_shm_get_config(ga3c_shm* shm, ga3c_shm_config* cfg) {
  if (!shm || !cfg) return fail(GA3C_H_EINVAL, "null argument");
  *cfg = shm->hdr()->cfg;
  return GA3C_H_OK;
}

int64_t ga3c_shm_state_offset(ga3c_shm* shm, int32_t agent) {
  if (!shm || agent < 0 || agent >= shm->hdr()->cfg.max_agents) return fail(GA3C_H_EINVAL, "bad agent id");
  return shm->hdr()->agents_off + agent * shm->hdr()->agent_stride;
}
int64_t ga3c_shm_agent_stride(ga3c_shm* shm) { return shm ? shm->hdr()->agent_stride : 0; }
int64_t ga3c_shm_rollout_offset(ga3c_shm* shm, int32_t slot) {
  if (!shm || slot < 0 || slot >= shm->hdr()->cfg.train_slots) return fail(GA3C_H_EINVAL, "bad slot id");
  return shm->hdr()->rollouts_off + slot * shm->hdr()->rollout_stride;
}
int64_t ga3c_shm_rollout_stride(ga3c_shm* shm) { return shm ? shm->hdr()->rollout_stride : 0; }

void* ga3c_pq_state_ptr(ga3c_shm* shm, int32_t agent) {
  if (!shm || agent < 0 || agent >= shm->hdr()->cfg.max_agents) return nullptr;
  return shm->base + shm->hdr()->agents_off + agent * shm->hdr()->agent_stride;
}

int ga3c_pq_submit(ga3c_shm* shm, int32_t agent) { return ga3c_pq_submit_flags(shm, agent, 0); }

int ga3c_pq_request_flags(ga3c_shm* shm, const uint32_t* ids, int32_t n, uint32_t* flags) {
  if (!shm || !ids || !flags || n < 0) return fail(GA3C_H_EINVAL, "bad argument");
  for (int i = 0; i < n; ++i) {
    if (ids[i] >= (uint32_t)shm->hdr()->cfg.max_agents) return fail(GA3C_H_EINVAL, "agent id %u out of range", ids[i]);
    flags[i] = shm->meta((int)ids[i])->req_flags;
  }
  return GA3C_H_OK;
}

int ga3c_pq_submit_flags(ga3c_shm* shm, int32_t agent, uint32_t flags) {
  if (!shm || agent < 0 || agent >= shm->hdr()->cfg.max_agents) return fail(GA3C_H_EINVAL, "bad agent id");
  Header* h = shm->hdr();
  if (h->closed.load(std::memory_order_acquire)) return GA3C_H_ECLOSED;
  AgentMeta* m = shm->meta(agent);
  if (m->req_seq != m->resp_seq.load(std::memory_order_acquire))
    return fail(GA3C_H_EINVAL, "agent %d already has a request in flight", agent);
  m->req_flags = flags;
  m->req_seq += 1;
  if (m->req_seq == 0) m->req_epoch += 1;
  std::atomic_thread_fence(std::memory_order_release);   // state bytes before the id becomes visible
  if (!ring_push(shm->base, &h->req, (uint32_t)agent, &h->closed)) {
    if (m->req_seq == 0) m->req_epoch -= 1;
    m->req_seq -= 1;
    return GA3C_H_ECLOSED;
  }
  return GA3C_H_OK;
}

static inline int64_t request_number(const AgentMeta* m) { return ((int64_t)m->req_epoch << 32) | (int64_t)m->req_seq; }

int ga3c_pq_request_seq(ga3c_shm* shm, int32_t agent, int64_t* seq) {
  if (!shm || !seq || agent < 0 || agent >= shm->hdr()->cfg.max_agents) return fail(GA3C_H_EINVAL, "bad argument");
  *seq = request_number(shm->meta(agent));
  return GA3C_H_OK;
}

int ga3c_pq_wait(ga3c_shm* shm, int32_t agent, float* p, float* v, int32_t timeout_ms) {
  if (!shm || !p || !v || agent < 0 || agent >= shm->hdr()->cfg.max_agents) return fail(GA3C_H_EINVAL, "bad argument");
  Header* h = shm->hdr();
  AgentMeta* m = shm->meta(agent);
  const uint32_t want = m->req_seq;
  const int64_t deadline = timeout_ms >= 0 ? now_ms() + timeout_ms : -1;
  // An answer that is already there, or arrives within spin_us, costs neither side a system call: the agent announces its
  // sleep in `waiting` (then looks once more), ga3c_pq_respond publishes the answer and wakes only an agent that has announced
  // it.  Both sides use sequentially consistent accesses, so one of them always sees the other's store.
  int slept = 0;
  const int32_t spin_us = h->spin_us.load(std::memory_order_relaxed);
  if (spin_us > 0 && m->resp_seq.load(std::memory_order_acquire) != want) {
    const int64_t until = now_ns() + (int64_t)spin_us * 1000;
    while (m->resp_seq.load(std::memory_order_acquire) != want && now_ns() < until) __builtin_ia32_pause();
  }
  for (;;) {
    uint32_t got = m->resp_seq.load(std::memory_order_acquire);
    if (got == want) break;
    if (h->closed.load(std::memory_order_acquire)) return GA3C_H_ECLOSED;
    int wait_ms = -1;
    if (deadline >= 0) {
      const int64_t left = deadline - now_ms();
      if (left <= 0) return GA3C_H_ETIMEOUT;
      wait_ms = (int)left;
    }
    m->waiting.store(1, std::memory_order_seq_cst);
    got = m->resp_seq.load(std::memory_order_seq_cst);
    if (got != want) { futex_wait(&m->resp_seq, got, wait_ms); slept = 1; }
    m->waiting.store(0, std::memory_order_relaxed);
  }
  {
    const int64_t age = now_ns() - (int64_t)m->answered_ns;      // both sides read CLOCK_MONOTONIC: comparable across processes
    if (age >= 0 && age < (int64_t)4000000000) {
      m->lat_sum_ns[slept] += (uint64_t)age;
      m->lat_count[slept] += 1;
      if ((uint32_t)age > m->lat_max_ns[slept]) m->lat_max_ns[slept] = (uint32_t)age;
    }
  }
  memcpy(p, m->p, (size_t)h->cfg.num_actions * sizeof(float));
  *v = m->v;
  return GA3C_H_OK;
}

int ga3c_pq_wake_latency(ga3c_shm* shm, int64_t* out6) {
  if (!shm || !out6) return fail(GA3C_H_EINVAL, "bad argument");
  int64_t r[6] = {0, 0, 0, 0, 0, 0};
  for (int a = 0; a < shm->hdr()->cfg.max_agents; ++a) {
    const AgentMeta* m = shm->meta(a);
    for (int k = 0; k < 2; ++k) {
      r[3 * k + 0] += m->lat_count[k];
      r[3 * k + 1] += (int64_t)m->lat_sum_ns[k];
      if ((int64_t)m->lat_max_ns[k] > r[3 * k + 2]) r[3 * k + 2] = m->lat_max_ns[k];
    }
  }
  memcpy(out6, r, sizeof r);
  return GA3C_H_OK;
}

int ga3c_pq_round_trip(ga3c_shm* shm, int32_t agent, const void* state, int32_t state_bytes, uint32_t flags, int32_t submit,
                       int32_t timeout_ms, double u, float* p, float* v, int32_t* action) {
  if (!shm || !p || !v || agent < 0 || agent >= shm->hdr()->cfg.max_agents) return fail(GA3C_H_EINVAL, "bad argument");
  if (submit) {
    if (state) {
      if (state_bytes < 0 || state_bytes > shm->hdr()->cfg.state_bytes) return fail(GA3C_H_EINVAL, "state of %d bytes does not fit the slot", state_bytes);
      memcpy(shm->base + shm->hdr()->agents_off + agent * shm->hdr()->agent_stride, state, (size_t)state_bytes);
    }
    const int rc = ga3c_pq_submit_flags(shm, agent, flags);
    if (rc != GA3C_H_OK) return rc;
  }
  const int rc = ga3c_pq_wait(shm, agent, p, v, timeout_ms);
  if (rc != GA3C_H_OK) return rc;
  if (action) *action = u >= 0.0 ? ga3c_select_action(p, shm->hdr()->cfg.num_actions, u) : -1;
  return GA3C_H_OK;
}

int ga3c_pq_set_spin(ga3c_shm* shm, int32_t spin_us) {
  if (!shm || spin_us < 0) return fail(GA3C_H_EINVAL, "bad argument");
  shm->hdr()->spin_us.store(spin_us, std::memory_order_relaxed);
  return GA3C_H_OK;
}

int ga3c_frame_queue_push(const uint32_t* in, const uint8_t* plane, uint32_t* out, int32_t n) {
  if (!in || !plane || !out || n < 0) return fail(GA3C_H_EINVAL, "bad argument");
  for (int32_t i = 0; i < n; ++i) out[i] = (in[i] >> 8) | ((uint32_t)plane[i] << 24);
  return GA3C_H_OK;
}

int32_t ga3c_select_action(const float* p, int32_t n, double u) {
  if (!p || n < 1) return 0;
  if (n > MAXA) n = MAXA;
  double cdf[MAXA];
  double acc = 0.0;
  for (int i = 0; i < n; ++i) {           // numpy: p converted to float64, cumsum = sequential adds
    acc += (double)p[i];
    cdf[i] = acc;
  }
  const double last = cdf[n - 1];
  int idx = 0;
  while (idx < n && !(u < cdf[idx] / last)) ++idx;     // searchsorted(side='right'): first index with cdf[idx] > u
  return idx < n ? idx : n - 1;
}

int ga3c_pq_agent_idle(ga3c_shm* shm, int32_t agent) {
  if (!shm || agent < 0 || agent >= shm->hdr()->cfg.max_agents) return fail(GA3C_H_EINVAL, "bad agent id");
  AgentMeta* m = shm->meta(agent);
  return m->req_seq == m->resp_seq.load(std::memory_order_acquire) ? 1 : 0;
}

int ga3c_pq_pop_batch(ga3c_shm* shm, uint32_t* ids, int32_t max_ids, int32_t timeout_ms) {
  if (!shm || !ids || max_ids < 1) return fail(GA3C_H_EINVAL, "bad argument");
  Header* h = shm->hdr();
  const int rc = ring_pop_wait(shm, &h->req, &ids[0], timeout_ms);
  if (rc == GA3C_H_ETIMEOUT) return 0;
  if (rc != GA3C_H_OK) return rc;
  int n = 1;
  while (n < max_ids && ring_try_pop(shm->base, &h->req, &ids[n])) ++n;
  // (the pushes of a burst wake ONE sleeping consumer: if this batch is full and more is queued, the next one is woken here)
  if (n == max_ids && h->req.waiters.load(std::memory_order_seq_cst) != 0 && ring_size(&h->req) != 0) futex_wake(&h->req.signal, 1);
  // optional linger (off by default: the reference drains without waiting, ThreadPredictor.py:54-55): keep collecting for
  // up to linger_us while fewer than linger_batch requests are in hand -- every forward pass has a fixed cost of tens
  // of microseconds, so a few more rows per pass can be worth a short wait
  const int32_t lus = h->linger_us.load(std::memory_order_relaxed);
  if (lus > 0) {
    int want = h->linger_batch.load(std::memory_order_relaxed);
    if (want > max_ids) want = max_ids;
    const int64_t until = now_ns() + (int64_t)lus * 1000;
    while (n < want && now_ns() < until && !h->closed.load(std::memory_order_acquire)) {
      if (ring_try_pop(shm->base, &h->req, &ids[n])) ++n;
      else sched_yield();
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return n;
}

int ga3c_pq_set_linger(ga3c_shm* shm, int32_t linger_us, int32_t min_batch) {
  if (!shm || linger_us < 0 || min_batch < 0) return fail(GA3C_H_EINVAL, "bad argument");
  shm->hdr()->linger_batch.store(min_batch, std::memory_order_relaxed);
  shm->hdr()->linger_us.store(linger_us, std::memory_order_relaxed);
  return GA3C_H_OK;
}

int ga3c_pq_respond(ga3c_shm* shm, const uint32_t* ids, int32_t n, const float* p, const float* v) {
  if (!shm || !ids || !p || !v || n < 0) return fail(GA3C_H_EINVAL, "bad argument");
  Header* h = shm->hdr();
  const int A = h->cfg.num_actions;
  for (int i = 0; i < n; ++i) {
    if (ids[i] >= (uint32_t)h->cfg.max_agents) return fail(GA3C_H_EINVAL, "agent id %u out of range", ids[i]);
    AgentMeta* m = shm->meta((int)ids[i]);
    memcpy(m->p, p + (size_t)i * A, (size_t)A * sizeof(float));
    m->v = v[i];
    m->answered_ns = (uint64_t)now_ns();
    m->resp_seq.store(m->req_seq, std::memory_order_seq_cst);
    if (m->waiting.load(std::memory_order_seq_cst) != 0) futex_wake(&m->resp_seq, 1);
  }
  return GA3C_H_OK;
}

int ga3c_pq_serve(ga3c_shm* shm, ga3c_predict_rows_fn predict, void* net, int32_t u8, int32_t max_batch,
                  int32_t slice_ms, ga3c_serve_stats* st) {
  if (!shm || !predict || !st || max_batch < 1 || slice_ms < 1) return fail(GA3C_H_EINVAL, "bad argument");
  name_this_thread("ga3c-predict");
  Header* h = shm->hdr();
  const int A = h->cfg.num_actions;
  std::vector<uint32_t> ids((size_t)max_batch);
  std::vector<int64_t> offs((size_t)max_batch);
  std::vector<float> p((size_t)max_batch * A), v((size_t)max_batch);
  const int64_t t_end = now_ns() + (int64_t)slice_ms * 1000000;
  for (;;) {
    const int64_t t0 = now_ns();
    const int64_t left_ms = (t_end - t0 + 999999) / 1000000;
    if (left_ms <= 0) return GA3C_H_OK;
    const int n = ga3c_pq_pop_batch(shm, ids.data(), max_batch, (int)left_ms);
    const int64_t t1 = now_ns();
    st->ns_pop += t1 - t0;
    if (n < 0) return n;
    if (n == 0) continue;
    for (int i = 0; i < n; ++i) {
      if (ids[i] >= (uint32_t)h->cfg.max_agents) return fail(GA3C_H_EINVAL, "agent id %u out of range", ids[i]);
      offs[i] = h->agents_off + (int64_t)ids[i] * h->agent_stride;
    }
    const int rc = predict(net, offs.data(), n, u8, p.data(), v.data(), nullptr);
    if (rc < 0) return fail(GA3C_H_ECALLBACK, "predict callback failed with %d on a batch of %d", rc, n);
    const int64_t t2 = now_ns();
    const int rr = ga3c_pq_respond(shm, ids.data(), n, p.data(), v.data());
    if (rr < 0) return rr;
    const int64_t t3 = now_ns();
    st->ns_predict += t2 - t1;
    st->ns_respond += t3 - t2;
    st->batches += 1;
    st->served += n;
    if (n > st->largest_batch) st->largest_batch = n;
  }
}

// The answers of a batch handed to a helper thread, so that the loop below goes straight on to the next batch: answering
// batch k itself, after it had popped and launched batch k+1, the loop woke k's agents ~19 us + their place in the
// batch after the results were there.  The helper sleeps on a futex between jobs (GA3C_RESPONDER_SPIN_US > 0: spins that
// long first -- no faster on a 16-core quota, and a core more); GA3C_RESPONDER=0 keeps the answers in the loop.  Measured
// (profiles/README.md): +2 % predictions/s with 64 agents, +4 % with 32 -- the loop's cycle is the GPU's latency either way,
// the agents' earlier requests wait in the queue instead.
// GA3C_RESPONDER: who wakes the agents of a finished batch.
//   0 (default)  the loop itself: in the pipelined loop after it has popped and begun the next batch (beside the GPU's work
//                on it; at once when nothing is queued), in the frames loop right after the batch
//   1            a helper thread of the call (round 3's default)
//   2            loop and helper share a batch of 8 rows or more (rows [0, n/2) the loop, [n/2, n) the helper)
//   3            the loop itself, always at once (before it pops the next batch)
// Measured on the placed engine (ga3c_amd/Placement.py; profiles/r04_engine_matrix.md), predictions/s with modes 0 / 1 / 2,
// same window within each group: 256 native agents, frame queue on the device 1,032 k / 949 k / 956 k (16 CPUs), 991 k / - /
// 915 k (64); 64 native 604 k / - / 641 k (16), 629 k / - / 628 k (64); 64 Python agents, device queue 456 k / 434 k / 443 k
// (16), 510 k / - / 493 k (64); 32 Python agents 290 k / 281 k / 284 k (16), states shipped 329 k / - / 302 k, mode 3 301 k
// (64).  The helper only pays while wake calls are dear (~2 us each on a cold core: unplaced, round 3); on warm cores a wake
// is ~1 us, the loop's own answers delay its next pop by less than a launch, and batches grow instead (92 rows against 54).
int responder_mode() {
  const char* he = getenv("GA3C_RESPONDER");
  return he ? atoi(he) : 0;
}

struct Responder {
  ga3c_shm* shm = nullptr;
  std::thread th;
  std::atomic<uint32_t> posted{0}, finished{0}, quit{0}, asleep{0};
  const uint32_t* ids = nullptr;
  const float* p = nullptr;
  const float* v = nullptr;
  int n = 0;
  std::atomic<int> rc{GA3C_H_OK};
  int64_t ns = 0;
  int spin_us = 0;
  void run() {
    name_this_thread("ga3c-respond");
    uint32_t seen = 0;
    for (;;) {
      const int64_t t_idle = now_ns();
      while (posted.load(std::memory_order_acquire) == seen) {
        if (quit.load(std::memory_order_acquire)) return;
        if (now_ns() - t_idle < (int64_t)spin_us * 1000) {
          __builtin_ia32_pause();
        } else {
          asleep.store(1, std::memory_order_seq_cst);
          if (posted.load(std::memory_order_seq_cst) == seen && !quit.load(std::memory_order_seq_cst)) futex_wait(&posted, seen, 2);
          asleep.store(0, std::memory_order_relaxed);
        }
      }
      seen = posted.load(std::memory_order_acquire);
      const int64_t t0 = now_ns();
      const int rr = ga3c_pq_respond(shm, ids, n, p, v);
      ns += now_ns() - t0;
      if (rr < 0) rc.store(rr, std::memory_order_relaxed);
      finished.store(seen, std::memory_order_release);
    }
  }
  void wait_idle() {
    while (finished.load(std::memory_order_acquire) != posted.load(std::memory_order_relaxed)) __builtin_ia32_pause();
  }
  void post(const uint32_t* i, int cnt, const float* pp, const float* vv) {
    wait_idle();                                             // (an answer takes a third of the loop's cycle: it is idle)
    ids = i; n = cnt; p = pp; v = vv;
    posted.fetch_add(1, std::memory_order_seq_cst);
    if (asleep.load(std::memory_order_seq_cst)) futex_wake(&posted, 1);
  }
  void stop() {
    wait_idle();
    quit.store(1, std::memory_order_seq_cst);
    futex_wake(&posted, 1);
    if (th.joinable()) th.join();
  }
};

static int serve_pipelined_common(ga3c_shm* shm, ga3c_predict_begin_fn begin, ga3c_predict_begin_cached_fn begin_cached,
                                  ga3c_predict_end_fn end, void* net, int32_t u8, int32_t max_batch, int32_t slice_ms,
                                  ga3c_serve_stats* st) {
  if (!shm || (!begin && !begin_cached) || !end || !st || max_batch < 1 || slice_ms < 1) return fail(GA3C_H_EINVAL, "bad argument");
  Header* h = shm->hdr();
  const int A = h->cfg.num_actions;
  // two batches in flight: `cur` is being computed while `prev` (results already fetched) is answered
  std::vector<uint32_t> ids[2] = {std::vector<uint32_t>((size_t)max_batch), std::vector<uint32_t>((size_t)max_batch)};
  std::vector<float> p[2] = {std::vector<float>((size_t)max_batch * A), std::vector<float>((size_t)max_batch * A)};
  std::vector<float> v[2] = {std::vector<float>((size_t)max_batch), std::vector<float>((size_t)max_batch)};
  std::vector<int64_t> offs((size_t)max_batch), seqs((size_t)max_batch);
  std::vector<int32_t> agents((size_t)max_batch);
  int cur = 0, n_prev = 0;
  const int64_t t_end = now_ns() + (int64_t)slice_ms * 1000000;
  name_this_thread("ga3c-predict");
  const int mode = responder_mode();
  const bool use_helper = mode == 1 || mode == 2;
  Responder helper;
  if (use_helper) {
    helper.shm = shm;
    if (const char* e = getenv("GA3C_RESPONDER_SPIN_US")) helper.spin_us = atoi(e);
    helper.th = std::thread([&helper] { helper.run(); });
  }
  struct StopHelper {                                        // every return below: nothing is held across slices
    Responder& r; ga3c_serve_stats* st; bool on;
    ~StopHelper() { if (on) { r.stop(); st->ns_respond += r.ns; } }
  } stop_helper{helper, st, use_helper};
  auto answer_prev = [&]() -> int {
    if (n_prev == 0) return GA3C_H_OK;
    const int64_t t0 = now_ns();
    const int rr = ga3c_pq_respond(shm, ids[1 - cur].data(), n_prev, p[1 - cur].data(), v[1 - cur].data());
    st->ns_respond += now_ns() - t0;
    n_prev = 0;
    return rr;
  };
  for (;;) {
    const int64_t t0 = now_ns();
    const int64_t left_ms = (t_end - t0 + 999999) / 1000000;
    if (left_ms <= 0) return answer_prev();                  // nothing is held across slices
    if (use_helper && helper.rc.load(std::memory_order_relaxed) < 0) return helper.rc.load();
    // with a batch waiting to be answered only requests that are ALREADY queued are taken; otherwise sleep for one
    const int n = ga3c_pq_pop_batch(shm, ids[cur].data(), max_batch, n_prev ? 0 : (int)left_ms);
    const int64_t t1 = now_ns();
    st->ns_pop += t1 - t0;
    if (n < 0 && n != GA3C_H_ETIMEOUT) {
      (void)answer_prev();
      return n;
    }
    int ticket = -1;
    if (n > 0) {
      for (int i = 0; i < n; ++i) {
        if (ids[cur][i] >= (uint32_t)h->cfg.max_agents) {
          (void)answer_prev();
          return fail(GA3C_H_EINVAL, "agent id %u out of range", ids[cur][i]);
        }
        offs[i] = h->agents_off + (int64_t)ids[cur][i] * h->agent_stride;
        if (begin_cached) {                                  // the row's name in the engine's state cache
          agents[i] = (int32_t)ids[cur][i];
          seqs[i] = request_number(shm->meta((int)ids[cur][i]));
        }
      }
      const int rc = begin_cached ? begin_cached(net, offs.data(), agents.data(), seqs.data(), n, u8, &ticket)
                                  : begin(net, offs.data(), n, u8, &ticket);
      if (rc < 0) {
        (void)answer_prev();
        return fail(GA3C_H_ECALLBACK, "predict callback (begin) failed with %d on a batch of %d", rc, n);
      }
    }
    const int64_t t2 = now_ns();
    st->ns_predict += t2 - t1;
    const int rr = answer_prev();                            // beside the GPU's work on `cur` (helper: nothing left to answer)
    if (n > 0) {
      const int64_t t3 = now_ns();
      const int rc = end(net, ticket, n, p[cur].data(), v[cur].data());
      st->ns_predict += now_ns() - t3;
      if (rc < 0) return fail(GA3C_H_ECALLBACK, "predict callback (end) failed with %d on a batch of %d", rc, n);
      st->batches += 1;
      st->served += n;
      if (n > st->largest_batch) st->largest_batch = n;
      if (mode == 3) {
        const int64_t t4 = now_ns();
        const int r3 = ga3c_pq_respond(shm, ids[cur].data(), n, p[cur].data(), v[cur].data());
        st->ns_respond += now_ns() - t4;
        if (r3 < 0) return r3;
      } else if (use_helper) {
        const int mine = (mode == 2 && n >= 8) ? n / 2 : 0;     // the buffers stay untouched until the helper has finished
        helper.post(ids[cur].data() + mine, n - mine, p[cur].data() + (size_t)mine * A, v[cur].data() + mine);
        if (mine) {
          const int64_t t4 = now_ns();
          const int r2 = ga3c_pq_respond(shm, ids[cur].data(), mine, p[cur].data(), v[cur].data());
          st->ns_respond += now_ns() - t4;
          if (r2 < 0) return r2;
        }
      } else {
        n_prev = n;
      }
      cur = 1 - cur;
    }
    if (rr < 0) {
      (void)answer_prev();
      return rr;
    }
  }
}

int ga3c_pq_serve_pipelined(ga3c_shm* shm, ga3c_predict_begin_fn begin, ga3c_predict_end_fn end, void* net, int32_t u8,
                            int32_t max_batch, int32_t slice_ms, ga3c_serve_stats* st) {
  if (!begin) return fail(GA3C_H_EINVAL, "bad argument");
  return serve_pipelined_common(shm, begin, nullptr, end, net, u8, max_batch, slice_ms, st);
}

int ga3c_pq_serve_pipelined_cached(ga3c_shm* shm, ga3c_predict_begin_cached_fn begin, ga3c_predict_end_fn end, void* net,
                                   int32_t u8, int32_t max_batch, int32_t slice_ms, ga3c_serve_stats* st) {
  if (!begin) return fail(GA3C_H_EINVAL, "bad argument");
  return serve_pipelined_common(shm, nullptr, begin, end, net, u8, max_batch, slice_ms, st);
}

int ga3c_pq_serve_frames(ga3c_shm* shm, ga3c_serve_frames_fn serve, void* net, int32_t max_batch, int32_t slice_ms,
                         ga3c_serve_stats* st) {
  if (!shm || !serve || !st || max_batch < 1 || slice_ms < 1) return fail(GA3C_H_EINVAL, "bad argument");
  Header* h = shm->hdr();
  const int A = h->cfg.num_actions;
  // two sets of result buffers: the helper thread answers batch k out of one while batch k+1 is served into the other
  std::vector<uint32_t> ids[2] = {std::vector<uint32_t>((size_t)max_batch), std::vector<uint32_t>((size_t)max_batch)};
  std::vector<float> p[2] = {std::vector<float>((size_t)max_batch * A, 0.f), std::vector<float>((size_t)max_batch * A, 0.f)};
  std::vector<float> v[2] = {std::vector<float>((size_t)max_batch, 0.f), std::vector<float>((size_t)max_batch, 0.f)};
  std::vector<uint32_t> flags((size_t)max_batch);
  std::vector<int32_t> agents((size_t)max_batch);
  std::vector<int64_t> offs((size_t)max_batch);
  const int64_t t_end = now_ns() + (int64_t)slice_ms * 1000000;
  name_this_thread("ga3c-predict");
  const int mode = responder_mode();
  const bool use_helper = mode == 1 || mode == 2;
  Responder helper;
  if (use_helper) {
    helper.shm = shm;
    if (const char* e = getenv("GA3C_RESPONDER_SPIN_US")) helper.spin_us = atoi(e);
    helper.th = std::thread([&helper] { helper.run(); });
  }
  struct StopHelper {
    Responder& r; ga3c_serve_stats* st; bool on;
    ~StopHelper() { if (on) { r.stop(); st->ns_respond += r.ns; } }
  } stop_helper{helper, st, use_helper};
  int cur = 0;
  for (;;) {
    const int64_t t0 = now_ns();
    const int64_t left_ms = (t_end - t0 + 999999) / 1000000;
    if (left_ms <= 0) return GA3C_H_OK;
    if (use_helper && helper.rc.load(std::memory_order_relaxed) < 0) return helper.rc.load();
    const int n = ga3c_pq_pop_batch(shm, ids[cur].data(), max_batch, (int)left_ms);
    const int64_t t1 = now_ns();
    st->ns_pop += t1 - t0;
    if (n < 0) return n;
    if (n == 0) continue;
    int predicted = 0;
    for (int i = 0; i < n; ++i) {
      if (ids[cur][i] >= (uint32_t)h->cfg.max_agents) return fail(GA3C_H_EINVAL, "agent id %u out of range", ids[cur][i]);
      offs[i] = h->agents_off + (int64_t)ids[cur][i] * h->agent_stride;
      agents[i] = (int32_t)ids[cur][i];
      flags[i] = shm->meta((int)ids[cur][i])->req_flags;
      predicted += (flags[i] & GA3C_REQ_NO_PREDICT) ? 0 : 1;
    }
    const int rc = serve(net, offs.data(), agents.data(), flags.data(), n, p[cur].data(), v[cur].data());
    if (rc < 0) return fail(GA3C_H_ECALLBACK, "serve callback failed with %d on a batch of %d", rc, n);
    const int64_t t2 = now_ns();
    st->ns_predict += t2 - t1;
    if (use_helper) {
      const int mine = (mode == 2 && n >= 8) ? n / 2 : 0;
      helper.post(ids[cur].data() + mine, n - mine, p[cur].data() + (size_t)mine * A, v[cur].data() + mine);   // (waits for the answers of the batch before)
      if (mine) {
        const int rr = ga3c_pq_respond(shm, ids[cur].data(), mine, p[cur].data(), v[cur].data());
        if (rr < 0) return rr;
        st->ns_respond += now_ns() - t2;
      }
      cur = 1 - cur;
    } else {
      const int rr = ga3c_pq_respond(shm, ids[cur].data(), n, p[cur].data(), v[cur].data());
      if (rr < 0) return rr;
      st->ns_respond += now_ns() - t2;
    }
    st->batches += 1;
    st->served += predicted;
    if (n > st->largest_batch) st->largest_batch = n;
  }
}

int ga3c_pq_serve_frames_pipelined(ga3c_shm* shm, ga3c_serve_frames_begin_fn begin, ga3c_serve_frames_end_fn end, void* net,
                                   int32_t max_batch, int32_t slice_ms, ga3c_serve_stats* st) {
  if (!shm || !begin || !end || !st || max_batch < 1 || slice_ms < 1) return fail(GA3C_H_EINVAL, "bad argument");
  Header* h = shm->hdr();
  const int A = h->cfg.num_actions;
  // `cur` is on the GPU while `prev` (results already fetched) is answered
  std::vector<uint32_t> ids[2] = {std::vector<uint32_t>((size_t)max_batch), std::vector<uint32_t>((size_t)max_batch)};
  std::vector<float> p[2] = {std::vector<float>((size_t)max_batch * A, 0.f), std::vector<float>((size_t)max_batch * A, 0.f)};
  std::vector<float> v[2] = {std::vector<float>((size_t)max_batch, 0.f), std::vector<float>((size_t)max_batch, 0.f)};
  std::vector<uint32_t> flags((size_t)max_batch);
  std::vector<int32_t> agents((size_t)max_batch);
  std::vector<int64_t> offs((size_t)max_batch);
  const int64_t t_end = now_ns() + (int64_t)slice_ms * 1000000;
  name_this_thread("ga3c-predict");
  const bool at_once = responder_mode() == 3;                // answer a batch before the next pop
  // The held batch is answered beside the next one only when that next one is worth it: requests for at least `deep` rows
  // are already queued (a quarter of a full batch; GA3C_PIPELINE_MIN_QUEUED).  With fewer the held answers go out first and
  // the loop then sleeps for requests -- otherwise a closed population of agents splits into twice as many, half as large
  // batches (256 native agents: 41 rows instead of 126), and what the overlap gives the fixed cost per batch takes back.
  int deep = max_batch / 4 > 1 ? max_batch / 4 : 1;
  if (const char* e = getenv("GA3C_PIPELINE_MIN_QUEUED")) deep = atoi(e);
  int cur = 0, n_prev = 0;
  auto answer_prev = [&]() -> int {
    if (n_prev == 0) return GA3C_H_OK;
    const int64_t t0 = now_ns();
    const int rr = ga3c_pq_respond(shm, ids[1 - cur].data(), n_prev, p[1 - cur].data(), v[1 - cur].data());
    st->ns_respond += now_ns() - t0;
    n_prev = 0;
    return rr;
  };
  for (;;) {
    const int64_t t0 = now_ns();
    const int64_t left_ms = (t_end - t0 + 999999) / 1000000;
    if (left_ms <= 0) return answer_prev();                  // nothing is held across slices
    if (n_prev && (int)ring_size(&h->req) < deep) {
      const int rr = answer_prev();
      if (rr < 0) return rr;
    }
    const int64_t t0b = now_ns();
    const int n = ga3c_pq_pop_batch(shm, ids[cur].data(), max_batch, n_prev ? 0 : (int)left_ms);
    const int64_t t1 = now_ns();
    st->ns_pop += t1 - t0b;
    if (n < 0 && n != GA3C_H_ETIMEOUT) {
      (void)answer_prev();
      return n;
    }
    int ticket = -1, predicted = 0;
    if (n > 0) {
      for (int i = 0; i < n; ++i) {
        if (ids[cur][i] >= (uint32_t)h->cfg.max_agents) {
          (void)answer_prev();
          return fail(GA3C_H_EINVAL, "agent id %u out of range", ids[cur][i]);
        }
        offs[i] = h->agents_off + (int64_t)ids[cur][i] * h->agent_stride;
        agents[i] = (int32_t)ids[cur][i];
        flags[i] = shm->meta((int)ids[cur][i])->req_flags;
        predicted += (flags[i] & GA3C_REQ_NO_PREDICT) ? 0 : 1;
      }
      const int rc = begin(net, offs.data(), agents.data(), flags.data(), n, &ticket);
      if (rc < 0) {
        (void)answer_prev();
        return fail(GA3C_H_ECALLBACK, "serve callback (begin) failed with %d on a batch of %d", rc, n);
      }
    }
    const int64_t t2 = now_ns();
    st->ns_predict += t2 - t1;
    const int rr = answer_prev();                            // beside the GPU's work on `cur`
    if (n > 0) {
      const int64_t t3 = now_ns();
      const int rc = end(net, ticket, flags.data(), n, p[cur].data(), v[cur].data());
      st->ns_predict += now_ns() - t3;
      if (rc < 0) return fail(GA3C_H_ECALLBACK, "serve callback (end) failed with %d on a batch of %d", rc, n);
      st->batches += 1;
      st->served += predicted;
      if (n > st->largest_batch) st->largest_batch = n;
      if (at_once) {
        const int64_t t4 = now_ns();
        const int r3 = ga3c_pq_respond(shm, ids[cur].data(), n, p[cur].data(), v[cur].data());
        st->ns_respond += now_ns() - t4;
        if (r3 < 0) return r3;
      } else {
        n_prev = n;
        cur = 1 - cur;
      }
    }
    if (rr < 0) {
      (void)answer_prev();
      return rr;
    }
  }
}

int ga3c_tq_acquire(ga3c_shm* shm, int32_t timeout_ms) {
  if (!shm) return fail(GA3C_H_EINVAL, "null argument");
  uint32_t slot = 0;
  const int rc = ring_pop_wait(shm, &shm->hdr()->freeq, &slot, timeout_ms, true);
  return rc == GA3C_H_OK ? (int)slot : rc;
}

void* ga3c_tq_states(ga3c_shm* shm, int32_t slot) {
  if (!shm || slot < 0 || slot >= shm->hdr()->cfg.train_slots) return nullptr;
  return shm->rollout(slot);
}
float* ga3c_tq_returns(ga3c_shm* shm, int32_t slot) {
  if (!shm || slot < 0 || slot >= shm->hdr()->cfg.train_slots) return nullptr;
  return reinterpret_cast<float*>(shm->rollout(slot) + shm->hdr()->ro_returns_off);
}
int32_t* ga3c_tq_actions(ga3c_shm* shm, int32_t slot) {
  if (!shm || slot < 0 || slot >= shm->hdr()->cfg.train_slots) return nullptr;
  return reinterpret_cast<int32_t*>(shm->rollout(slot) + shm->hdr()->ro_actions_off);
}

int ga3c_tq_commit(ga3c_shm* shm, int32_t slot, int32_t rows) {
  if (!shm || slot < 0 || slot >= shm->hdr()->cfg.train_slots) return fail(GA3C_H_EINVAL, "bad slot id");
  if (rows < 1 || rows > shm->hdr()->cfg.train_rows) return fail(GA3C_H_EINVAL, "rows %d outside [1,%d]", rows, shm->hdr()->cfg.train_rows);
  *reinterpret_cast<int32_t*>(shm->rollout(slot) + shm->hdr()->ro_rows_off) = rows;
  std::atomic_thread_fence(std::memory_order_release);
  if (!ring_push(shm->base, &shm->hdr()->readyq, (uint32_t)slot, &shm->hdr()->closed)) return GA3C_H_ECLOSED;
  return GA3C_H_OK;
}

int ga3c_tq_pop(ga3c_shm* shm, int32_t timeout_ms) {
  if (!shm) return fail(GA3C_H_EINVAL, "null argument");
  uint32_t slot = 0;
  const int rc = ring_pop_wait(shm, &shm->hdr()->readyq, &slot, timeout_ms);
  std::atomic_thread_fence(std::memory_order_acquire);
  return rc == GA3C_H_OK ? (int)slot : rc;
}

int ga3c_tq_collect(ga3c_shm* shm, int32_t min_rows, int32_t timeout_ms, int32_t hold_timeout_ms, int32_t* rows,
                    int32_t* n_slots, int32_t* slots, int64_t* row_offsets, float* returns, int32_t* actions,
                    int32_t cap_rows, int32_t cap_slots, int64_t* row_seq, int32_t* row_agent) {
  if (!shm || !rows || !n_slots || !slots || !row_offsets || !returns || !actions) return fail(GA3C_H_EINVAL, "null argument");
  name_this_thread("ga3c-train");
  Header* h = shm->hdr();
  if (*rows < 0 || *n_slots < 0 || min_rows < 0) return fail(GA3C_H_EINVAL, "bad batch state");
  const int64_t row_bytes = h->cfg.rollout_row_bytes ? h->cfg.rollout_row_bytes : h->cfg.state_bytes;
  const bool names = row_seq != nullptr || row_agent != nullptr;     // rows only NAME states kept on the device
  if (names && (!row_seq || !row_agent || h->cfg.rollout_row_bytes != 16))
    return fail(GA3C_H_EINVAL, "row names need both arrays and 16-byte rollout rows");
  while (*rows <= min_rows) {
    if (*rows + h->cfg.train_rows > cap_rows || *n_slots >= cap_slots) return fail(GA3C_H_EINVAL, "batch arrays too small");
    // the caller keeps the slots until the GPU has read them: if the agents have none left and nothing is queued, it
    // has to give some back first (ThreadTrainer's spill rule)
    if (*n_slots > 0 && ring_size(&h->freeq) == 0 && ring_size(&h->readyq) == 0) return GA3C_H_ESTARVED;
    uint32_t slot = 0;
    const int rc = ring_pop_wait(shm, &h->readyq, &slot, *n_slots > 0 ? hold_timeout_ms : timeout_ms);
    if (rc != GA3C_H_OK) return rc;
    std::atomic_thread_fence(std::memory_order_acquire);
    const char* ro = shm->rollout((int)slot);
    const int32_t n = *reinterpret_cast<const int32_t*>(ro + h->ro_rows_off);
    const int64_t off0 = (int64_t)(ro - shm->base);
    for (int32_t i = 0; i < n; ++i) row_offsets[*rows + i] = off0 + i * row_bytes;
    memcpy(returns + *rows, ro + h->ro_returns_off, (size_t)n * sizeof(float));
    memcpy(actions + *rows, ro + h->ro_actions_off, (size_t)n * sizeof(int32_t));
    if (names) {                                             // (plane sequence number i64, agent id i32) per row; the slot is done
      for (int32_t i = 0; i < n; ++i) {
        memcpy(&row_seq[*rows + i], ro + i * row_bytes, 8);
        memcpy(&row_agent[*rows + i], ro + i * row_bytes + 8, 4);
      }
      const int rr = ga3c_tq_release(shm, (int32_t)slot);
      if (rr != GA3C_H_OK) return rr;
    } else {
      slots[*n_slots] = (int32_t)slot;
      *n_slots += 1;
    }
    *rows += n;
  }
  return GA3C_H_OK;
}

int ga3c_tq_release_many(ga3c_shm* shm, const int32_t* slots, int32_t n) {
  if (!shm || (!slots && n > 0) || n < 0) return fail(GA3C_H_EINVAL, "bad argument");
  for (int32_t i = 0; i < n; ++i) {
    const int rc = ga3c_tq_release(shm, slots[i]);
    if (rc != GA3C_H_OK) return rc;
  }
  return GA3C_H_OK;
}

int ga3c_tq_rows(ga3c_shm* shm, int32_t slot) {
  if (!shm || slot < 0 || slot >= shm->hdr()->cfg.train_slots) return fail(GA3C_H_EINVAL, "bad slot id");
  return *reinterpret_cast<int32_t*>(shm->rollout(slot) + shm->hdr()->ro_rows_off);
}

int ga3c_tq_release(ga3c_shm* shm, int32_t slot) {
  if (!shm || slot < 0 || slot >= shm->hdr()->cfg.train_slots) return fail(GA3C_H_EINVAL, "bad slot id");
  if (!ring_push(shm->base, &shm->hdr()->freeq, (uint32_t)slot, &shm->hdr()->closed)) return GA3C_H_ECLOSED;
  return GA3C_H_OK;
}

int ga3c_tq_ready_count(ga3c_shm* shm) {
  if (!shm) return fail(GA3C_H_EINVAL, "null argument");
  return (int)ring_size(&shm->hdr()->readyq);
}

int ga3c_tq_free_count(ga3c_shm* shm) {
  if (!shm) return fail(GA3C_H_EINVAL, "null argument");
  return (int)ring_size(&shm->hdr()->freeq);
}

}  // extern "C"
