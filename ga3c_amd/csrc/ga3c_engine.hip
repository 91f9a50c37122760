// ga3c_engine.hip -- C ABI (include/ga3c_abi.h) over the gfx950 kernels in ga3c_kernels.hpp.
//
// Replaces the TensorFlow session behind the reference's Network object
// (/root/reference/ga3c/NetworkVP.py:48-59,248-257).  HBM layout per net:
//   theta[2]   double-buffered flat f32 weight arena (TF variable order/layout), flipped by each
//              optimizer step so concurrent predictions never see a half-applied update;
//   grad, ms, mom   flat arenas in the same layout (grad is what RCCL all-reduces);
//   per lane   x[B,84,84,4]  n1[B,21,21,16]  n2[B,11,11,32]  part[KS,B,256]  d1[B,256]  z,p[B,A]  v[B]
//   train lane additionally  y_r, a, dz, dv, lossrow[B,3], dd1[B,256], dn2, dn1, slab2, slab1.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <atomic>
#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ga3c_abi.h"
#include "ga3c_kernels.hpp"
#include "ga3c_checkpoint.hpp"
#include "ga3c_frontend.hpp"
#include "ga3c_resample.hpp"

using namespace ga3c;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      (void)hipGetLastError(); /* reported here: it must not surface again behind a later launch */ \
      return fail(GA3C_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    }                                                                                             \
  } while (0)
#define NCCLCHK(expr)                                                                             \
  do {                                                                                            \
    ncclResult_t _e = (expr);                                                                     \
    if (_e != ncclSuccess) return fail(GA3C_ERCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_e), \
                                       __FILE__, __LINE__);                                       \
  } while (0)
#define CHK(expr)              \
  do {                         \
    int _r = (expr);           \
    if (_r != GA3C_OK) return _r; \
  } while (0)

// HIP streams of the process, kept for its lifetime and lent to the networks that live in it.  The runtime multiplexes the
// streams of a process onto four hardware queues per priority and gives a new stream whichever queue it counts least used;
// those counts do not go back to where they were when streams are destroyed, so where the streams of a SECOND network of the
// process land depends on the networks before it.  Measured with bench.py (round 4): behind the kernel-timing, Hogwild and
// data-parallel networks of its earlier legs the engine legs ran at 261 / 290 / 295 k predictions/s (the server's two
// prediction lanes no longer on queues of their own) against 320 / 517 / 516 k when the earlier networks happened to create an
// even number of streams, and against 318 k for the same server as the process's first network.  Streams are therefore
// created once, in a fresh process's order, and re-used: a network takes the first free stream of the priority it asks for
// and hands it back when it is destroyed (networks alive at the same time -- GA3C_mixed.py -- never share one).
struct PooledStream { hipStream_t st; int device; bool high; bool busy; };
std::mutex g_stream_mu;
std::vector<PooledStream> g_streams;

hipError_t stream_take(int device, bool high_priority, hipStream_t* out) {
  std::lock_guard<std::mutex> g(g_stream_mu);
  for (PooledStream& p : g_streams)
    if (!p.busy && p.device == device && p.high == high_priority) {
      p.busy = true;
      *out = p.st;
      return hipSuccess;
    }
  hipStream_t st = nullptr;
  hipError_t e;
  if (high_priority) {
    int lo = 0, hi = 0;
    e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
  } else {
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  }
  if (e != hipSuccess) return e;
  g_streams.push_back({st, device, high_priority, true});
  *out = st;
  return hipSuccess;
}

void stream_give_back(hipStream_t st) {
  if (!st) return;
  (void)hipStreamSynchronize(st);                    // nothing of the old owner is left on it
  std::lock_guard<std::mutex> g(g_stream_mu);
  for (PooledStream& p : g_streams)
    if (p.st == st) {
      p.busy = false;
      return;
    }
  (void)hipStreamDestroy(st);                         // not one of ours
}

// rocTX ranges around the engine's calls (tracing): `rocprofv3 --marker-trace` shows ga3c.predict /
// ga3c.train.stage / ga3c.train.step on the host timeline beside the kernels; a no-op without a profiler attached
struct TraceRange {
  explicit TraceRange(const char* name) { roctxRangePushA(name); }
  ~TraceRange() { roctxRangePop(); }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Fwd {   // forward workspace of one lane (device pointers)
  float *x = nullptr, *n1 = nullptr, *n2 = nullptr, *part = nullptr, *d1 = nullptr, *z = nullptr, *p = nullptr,
        *v = nullptr;
  uint8_t* xu8 = nullptr;
  bool x_u8 = false;   // the staged batch lives in xu8 (uint8 frames) and the conv kernels convert while reading
  // prediction intake fused into the conv stack: sample b lies src_off[b] bytes behind src_base (nullptr: dense batch)
  const uint8_t* src_base = nullptr;
  const int64_t* src_off = nullptr;
  const int64_t* src_off_host = nullptr;   // the same array as the host sees it (pinned: same address; kept apart for clarity)
  const int64_t* cache_dst = nullptr;      // per row: where in the state cache the conv stack stores the uint8 state it stages
};

struct Lane {
  std::mutex mu;
  hipStream_t st = nullptr;   // lanes beyond net->lane_streams borrow the stream of lane (index % lane_streams): see ga3c_net_create
  bool owns_st = true;
  bool shared_st = false;     // another lane enqueues on `st` too: completion is waited for on `done`, not on the stream
  int sidx = 0;               // which of the prediction streams `st` is (index into ga3c_net::stream_busy)
  std::atomic<bool> begun{false};   // taken by ga3c_net_predict_gather_begin, to be given back by _end
  int frames_want = 0;              // ga3c_net_serve_frames_begin: predictions enqueued for the batch _end will hand out
  int64_t* cache_dst = nullptr;     // pinned [maxB]: byte offsets into the state cache of the batch begun with _begin_cached
  bool cache_on = false;
  hipEvent_t done = nullptr;
  hipEvent_t tm0 = nullptr, tm1 = nullptr;   // timing events of ga3c_net_time_predict_lanes
  uint64_t staged_gen = 0;    // ga3c_net_time_predict_lanes: which resident batch this lane's x holds a copy of (0: none)
  int staged_rows = 0;
  bool staged_u8 = false;
  Fwd f;
  float* h_in = nullptr;    // pinned staging, max_batch states
  float* h_out = nullptr;   // pinned staging, p|v|z
  int64_t* h_off = nullptr; // pinned: per-row byte offsets, read in place by the gather kernels
  bool dirty[3] = {false, false, false};   // a step of this lane has read theta[i] since it was last written (guarded by wmu); `done` is behind it
  // captured prediction steps, one executable graph per (batch, weight buffer, intake mode, output target): every
  // pointer a step touches is fixed for the lane's lifetime, so a step is replayed with ONE launch call
  std::unordered_map<int64_t, hipGraphExec_t> graphs;
};

enum StepMode { STEP_RESIDENT = 0, STEP_GATHER_U8 = 1, STEP_GATHER_F32 = 2, STEP_QUEUES = 3 };

// Input side of a train step: the batch (states as f32 or uint8, returns, one-hot actions) and the pinned arrays it is
// staged through.  A train lane has TWO of them and a stream of its own for staging (`gst`): while one trainer thread's step
// runs on the train stream, the next thread gathers ITS batch out of the transport (3.7 MB over PCIe, ~100 us) into the other
// intake -- Server.py:132-134 runs NT trainer threads against one model for exactly this overlap.  Steps themselves stay
// strictly ordered by the lane's mutex (synchronous, reproducible).  Lock order: intake, then lane.
struct Intake {
  std::mutex mu;
  float* x = nullptr;
  uint8_t* xu8 = nullptr;
  float *yr = nullptr, *act = nullptr;
  float* h_in = nullptr;    // pinned: x | y_r | a
  int64_t* h_off = nullptr; // pinned: per-row byte offsets / plane sequence numbers, read in place by the gather kernels
  hipEvent_t ready = nullptr;   // recorded on the staging stream behind the batch
  hipEvent_t done = nullptr;    // recorded on the train stream behind the step that trained this batch
  float* losses = nullptr;      // pinned: the three loss sums of that step, written by the kernel that completes them
  int wbuf = -1;                // the weight buffer that step wrote (-1: none, or updated in place)
  uint64_t wbuf_seq = 0;        // ... and the step number it carried then (ga3c_net::wseq)
  bool x_u8 = false;
};

// where a batch is being staged: an intake's buffers (or the lane's currently bound ones) and the stream to stage on
struct Stage {
  float* x; uint8_t* xu8; float* yr; float* act; float* h_in; int64_t* h_off;
  hipStream_t st;
  bool x_u8;
};

struct TrainLane {
  std::mutex mu;
  hipStream_t st = nullptr;
  hipStream_t gst = nullptr;   // staging stream of the intakes
  Intake in[2];
  std::atomic<unsigned> in_rr{0};
  Fwd f;                    // f.x / f.xu8 / f.x_u8, yr, act, h_in, h_off below: the buffers of the intake bound last (bind_intake)
  float *yr = nullptr, *act = nullptr, *dz = nullptr, *dv = nullptr, *lossrow = nullptr, *dd1 = nullptr,
        *dn2 = nullptr, *dn1 = nullptr, *slab2 = nullptr, *slab1 = nullptr, *losses = nullptr, *scales = nullptr;
  float* h_in = nullptr;    // pinned: x | y_r | a
  float* h_out = nullptr;   // pinned: p | v | z | losses
  int64_t* h_off = nullptr;
  float* grad = nullptr;    // this lane's gradient arena (lane 0: net->grad, the buffer RCCL all-reduces)
  bool owns_grad = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool exchanged = false;   // the gradients in `grad` have been all-reduced by the backward pass itself (overlapped exchange)
  bool stepped = false;     // the backward pass has applied RMSProp itself (FusedUpd) into theta[stepped_other]
  int stepped_other = 0;
  int wrote = -1;           // the weight buffer the step enqueued last on this lane wrote (publish_other_buffer)
  uint64_t wrote_seq = 0;   // ... and that step's number
};

// K slices of dense1_fwd.  Its grid is (row blocks) x (2 column halves) x (slices), every slice a partial slab that
// `heads` reads back; the 242 K steps are cut as evenly as integers allow, so the count need not divide 242.  Measured
// on the MI355X (profiles/README.md): 16 slices give 256 / 512 / 512 workgroups at batch 128 / 256 / 512 -- whole rounds
// on the 256 CUs -- and beat the former 22 / 22 / 11 by 5 / 8 / 20 %; at batch 1024 8 slices (512 workgroups) beat 11.
int dense_ks(int B) {
  if (B > 128 && B <= 176) return 22;   // 9-11 row blocks: 16 slices would leave each XCD 36-44 workgroups for its 32 CUs
  return B <= 512 ? 16 : (B <= 1024 ? 8 : 4);
}
// conv_stack_fwd puts one 1024-thread workgroup (149.5 KB of LDS) on a CU, two per sample: up to 128 samples are one
// round on the 256 CUs, sample 129 starts a second one (11.3 us at batch 128, 19.9 us at 132), so above 128 the
// two-kernel form takes over (train steps/s at 132 rows: 10.3 k fused, 10.8 k split).
constexpr int FUSED_CONV_MAX_B = 128;
// 16-row tiles per wave in dense1_dx (weight fragments are reused across them).  Measured, dense1_bwd per launch:
// batch 256: 30.4 / 23.3 / 26.3 us with 4 / 2 / 1 tiles; batch 512: 39.6 / 41.4 / 46.3; batch 1024: 71.6 / 76.3 / 87.0.
int dense_dx_mt(int B) { return B > 384 ? 4 : (B > 192 ? 2 : 1); }
}  // namespace

// Frame front-end state (ga3c_net_frames_*): resample tables, one 4-deep frame queue per agent, staging.
struct Frames {
  bool on = false;
  int maxA = 0, H = 0, W = 0, C = 0, hks = 0, vks = 0;
  size_t frame_bytes = 0, lds = 0;
  int32_t* d_tab = nullptr;            // hb | hk | vb | vk
  const int32_t *hb = nullptr, *hk = nullptr, *vb = nullptr, *vk = nullptr;
  uint32_t* stacks = nullptr;          // [maxA][84*84] words = [84,84,4] uint8 HWC states
  uint8_t* d_rgb = nullptr;            // [maxA] resident frames (ga3c_net_frames_upload / timing)
  uint8_t* h_rgb = nullptr;            // pinned staging for pageable callers (read by the kernel in place)
  uint8_t* d_planes = nullptr;
  uint8_t* h_planes = nullptr;         // pinned
  int32_t* h_agents = nullptr;         // pinned, read by the kernel in place
  uint8_t* h_reset = nullptr;          // pinned
  int64_t* h_src = nullptr;            // pinned: per-frame source offsets (frames_push_offsets)
  int32_t* h_slot = nullptr;           // pinned: history slot of each pushed plane
  int hist = 0;                        // planes of history kept per agent (0: none)
  uint8_t* ring = nullptr;             // [maxA][hist][84*84]
  std::vector<int64_t> pushed;         // planes pushed per agent so far = sequence number of the next plane
  std::vector<int> filled;             // host mirror of each queue's depth (0..4)
  hipStream_t st = nullptr;
  std::mutex mu;
};

// Host threads that drive the prediction lanes in ga3c_net_time_predict_lanes, one per lane, alive as long as the net: a
// K-step block at the driver's K = 20 is ~0.3 ms of GPU work, and starting + joining a std::thread per lane inside every
// timed block cost ~0.13 ms of it (round-2 verdict).  A block is posted under `mu`; the threads then meet the caller in a
// spin rendezvous (ready -> go) so that the clock starts with every lane about to launch, and report through `done`.
struct LaneDrivers {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t seq = 0;                    // number of the block posted last (guarded by mu)
  std::atomic<uint64_t> seq_hint{0};   // the same number, readable without the mutex: a driver polls it for a while before it sleeps
  bool quit = false;
  int batch = 0, iters = 0, nlanes = 0, idx = 0;
  std::atomic<int> ready{0}, done{0};
  std::atomic<uint64_t> go{0};
  std::vector<int> rcs;
  std::vector<std::string> errs;
};

struct ga3c_net {
  ga3c_net_config cfg;
  int A = 0;
  int maxB = 0;
  int64_t n = 0;   // arena floats
  static constexpr int NBUF = 3;
  float* theta[NBUF] = {nullptr, nullptr, nullptr};
  float* theta_pk[NBUF] = {nullptr, nullptr, nullptr};   // dense1/w of theta[i] in dense1_fwd's fragment order (+ the packed conv12/w)
  // Triple-buffered weights (round 3; two buffers before).  `latest` is the buffer the newest ENQUEUED optimizer step wrote
  // (train-stream order; everything on the train side reads it); `cur` is the buffer prediction lanes read: it only ever
  // advances to a buffer whose step has FINISHED on the GPU, so a prediction never queues behind a train step in flight (it
  // reads weights one or two steps old instead -- the reference's predictors read weights in the middle of an update).
  // A step reads theta[latest] and writes a buffer that is neither `latest` nor `cur`: with three buffers there always is
  // one, also while the previous step is still in flight -- which is what lets two trainer threads enqueue their steps back
  // to back (with two buffers the second step had to move the predictions onto weights still being written, and every
  // prediction issued meanwhile waited for that step).
  std::atomic<int> cur{0};
  int latest = 0;
  uint64_t wseq[NBUF] = {0, 0, 0};      // number of the step that wrote theta[i] (guarded by wmu): newer = larger
  uint64_t wcount = 0;
  bool event_valid[NBUF] = {false, false, false};   // theta_ready[i] was recorded behind the step that wrote theta[i]
  std::atomic<uint64_t> pred_seq{0};    // prediction steps launched so far; the optimizer records theta_ready only while
  uint64_t pred_seen = 0;               // predictions are arriving (an event per step costs a train-only loop ~3 us)
  float *grad = nullptr, *ms = nullptr, *mom = nullptr;
  hipEvent_t theta_ready[NBUF] = {nullptr, nullptr, nullptr};   // recorded on the train stream behind the step that wrote theta[i]
  std::mutex ready_mu;
  std::shared_mutex wmu;   // shared: a forward pass picking/reading theta[cur]; unique: the optimizer flip
  std::vector<Lane*> lanes;
  std::atomic<unsigned> rr{0};
  TrainLane tr;                        // train lane 0 (the only one in synchronous mode)
  std::vector<TrainLane*> xtr;         // extra train lanes: Hogwild mode (cfg.train_lanes >= 2), as the reference's NT threads
  bool hogwild = false;
  std::atomic<unsigned> trr{0};
  std::atomic<int64_t> step{0};
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  // data-parallel exchange overlapped with the backward pass: dense1/w and everything behind it in the arena (98.8 % of
  // the gradient) is final after the first backward kernel and is all-reduced on `cst` while the conv gradients are
  // still being computed on the train stream; the 49 KB in front follow when they are done
  hipStream_t cst = nullptr;
  hipEvent_t ev_tail_ready = nullptr, ev_head_ready = nullptr, ev_comm_done = nullptr;
  bool head_on_train_stream = true;    // GA3C_COMM_HEAD_INLINE=0: the conv gradients' exchange hops to the comm stream too (round 3)
  bool comm_overlap = true;            // GA3C_COMM_OVERLAP=0: one blocking all-reduce of the whole arena behind the backward pass
  bool fused_conv = true;              // conv1+conv2 in one launch (GA3C_SPLIT_CONV=1 selects the two-kernel form)
  bool fused_update = true;            // single-GPU train steps: RMSProp applied by the kernels that complete each gradient
                                       // element, no optimizer launch (GA3C_FUSED_UPDATE=0: the rmsprop kernel)
  int c2dw_occ = 3;                    // conv2_dw up to 256 rows (one sample per workgroup): workgroups per CU its registers are cut for
                                       // (GA3C_C2DW_OCC=2: the 205-VGPR form with the next sample's loads in flight, which larger batches run;
                                       // measured 4.5 / 6.0 / 7.5 / 8.0 / 10.2 us against 4.6 / 6.2 / 9.0 / 9.3 / 11.2 at 64 / 128 / 132 / 192 / 256 rows)
  bool conv_bwd_fused = true;          // conv2_dw + conv2_dx + conv1_dw in one launch (GA3C_CONV_BWD=0: conv2_dx, then the two weight gradients)
  bool c2f_quarter = true;             // conv2_fwd at 129 .. 192 rows: four tiles per workgroup instead of eight (GA3C_C2F_QUARTER=0)
  int conv_bwd_min = 97;               // ... from this many rows on (GA3C_CONV_BWD_MIN).  Below, the split launches are faster since round 4
                                       // (uint8 train step 44.7 / 45.6 / 48.7 / 50.3 / 55.6 us against 49.3 / 50.1 / 51.5 / 52.9 / 56.4 at
                                       // 5 / 16 / 32 / 48 / 80 rows; 58.2 against 57.4 at 100): a small grid of the one-workgroup-per-CU kernel
                                       // leaves most of the chip idle, the split kernels spread the same rows over more workgroups
  // State cache (ga3c_net_state_cache_config): the uint8 states the prediction steps read out of the transport, kept in HBM
  // in a ring of `depth` per agent, slot = request number % depth -- a train batch then names its rows (agent, request
  // number) and is gathered HBM to HBM instead of crossing PCIe a second time.
  uint8_t* cache_ring = nullptr;
  int cache_agents = 0, cache_depth = 0;
  std::mutex cache_mu;
  std::vector<int64_t> cache_newest;   // per agent: the newest request number stored (-1: none)
  std::vector<int64_t> cache_tags;     // [agent][slot]: the request number the slot really holds (-1: nothing yet)
  bool frames_in_line = true;          // GA3C_FRAMES_IN_LINE=0: rows out of the plane history are staged on the staging stream, beside a step
  bool stop_events = true;             // GA3C_STOP_EVENTS=0: a prediction step's completion event is a hipEventRecord of its own
  bool offsets_in_args = true;         // GA3C_OFFSETS_IN_ARGS=0: the conv stack reads a scattered batch's offsets out of pinned host memory
  bool time_predictions = false;       // GA3C_TIME_PREDICTIONS=1: timing events around every prediction step (GA3C_STAT_PREDICT_GPU_NS)
  int c1dw_blocks = 0;                 // conv1_dw workgroups = partial slabs of the split path; 0: conv1_dw_blocks() picks (GA3C_C1DW_BLOCKS)
  bool dw_pair = true;                 // split path up to 256 rows: conv2_dw and conv1_dw in one launch behind conv2_dx (GA3C_DW_PAIR=0: two)
  int wd_blocks_first = 0;             // ... at the front (1) or at the back (0) of that grid (GA3C_WD_BLOCKS_FIRST)
  bool wd_step_in_conv2_dx = true;     // beyond 128 rows (split conv backward): dense1/w stepped by workgroups of their own in conv2_dx's
                                       // launch instead of in dense1_bwd_tile's epilogue (GA3C_WD_STEP_IN_CONV2_DX=0: the epilogue)
  int wd_step_in_conv_bwd = 1;         // fused update: dense1/w stepped inside conv_bwd (GA3C_WD_STEP_IN_CONV_BWD: 0 never -- in
                                       // dense1_bwd_tile's epilogue --, 1 when conv_bwd's grid covers the 242 row groups, 2 always)
                                       // Measured: 29 us against 5.7 + 5.8 us as two launches (profiles/README.md) -- kept for the record, off
  int d1f_frag_lanes = 2;              // prediction steps of >= 64 rows use the register-fragment dense1 (no LDS: it shares a CU
                                       // with another lane's conv stack, which the 148 KB tile kernel cannot) while at least
                                       // this many prediction calls are in flight (GA3C_D1F_FRAG_LANES; 0 = never).  Same bits.
                                       // Measured, 1 / 2 / 3 / 4 lanes at 128 rows: threshold 3 -> 5.80 / 7.51 / 9.07 / 10.53 M
                                       // predictions/s, 2 -> 5.81 / 7.79 / 9.01 / 10.60, 1 -> 5.55 / 7.78 / 8.98 / 10.59
  std::atomic<int> predict_inflight{0};
  bool d1f_tile = true;                // LDS-tiled dense1 forward where its grid is one round (GA3C_D1F_TILE=0: never)
  bool d1b_tail = true;                // dense1_bwd_tile at 129 .. 133 rows: the rows past the first chunk wait in the LDS a chunk leaves free
                                       // (D1B_TAIL_ROWS; GA3C_D1B_TAIL=0: a second round of staging loads, as for any larger batch)
  int d1b_tile_max = 1 << 30;          // largest batch that takes the LDS-tiled dense1 backward (GA3C_D1B_TILE_MAX overrides)
  bool graphs = false;                 // GA3C_GRAPHS=1: prediction steps replayed as hipGraphs.  Off by default: on ROCm 7.2 a
                                       // 4-kernel graph launch costs ~6 us MORE per step than four plain launches and
                                       // two lanes lose 10 % of their overlap (profiles/README.md, round 1)
  void* reg_host = nullptr;            // HIP-registered host segment (the shm transport) ...
  uint8_t* reg_dev = nullptr;          // ... and the device-side address of its first byte
  int64_t reg_bytes = 0;
  Frames fr;
  TensorTable tt;
  LaneDrivers drv;
  float lanes_gpu_ms = 0.f;            // GPU-side span of the last ga3c_net_time_predict_lanes block (first start event .. last end event)
  int lanes_gpu_n = 0;
  int gather_max_blocks = 32;          // workgroups of the PCIe gather (GA3C_GATHER_BLOCKS; ga3c_kernels.hpp: gather_rows_kernel)
  int lane_streams = 3;                // HIP streams the prediction lanes are spread over (GA3C_LANE_STREAMS)
  std::atomic<uint64_t> resident_gen{1};   // moves whenever the train lane's resident batch is (re)written
  std::atomic<int> stream_busy[16];    // lanes at work per prediction stream: take_lane prefers a lane whose stream is idle
  // where the engine's calls spend their time (ga3c_net_stats): nanoseconds / counts, relaxed atomics
  std::atomic<int64_t> stat[GA3C_STAT_COUNT];
};

namespace {

inline void stat_add(ga3c_net* net, int which, int64_t v) { net->stat[which].fetch_add(v, std::memory_order_relaxed); }

int dmalloc(float** p, size_t floats) {
  HIPCHK(hipMalloc((void**)p, floats * sizeof(float)));
  HIPCHK(hipMemset(*p, 0, floats * sizeof(float)));
  return GA3C_OK;
}

int alloc_fwd(Fwd& f, int maxB, int A) {
  size_t part = 0;
  for (int b : {maxB < 256 ? maxB : 256, maxB < 1024 ? maxB : 1024, maxB}) {
    const size_t need = (size_t)22 * b * HID;   // dense_ks() <= 22
    if (need > part) part = need;
  }
  CHK(dmalloc(&f.x, (size_t)maxB * XS));
  CHK(dmalloc(&f.n1, (size_t)maxB * N1S));
  CHK(dmalloc(&f.n2, (size_t)maxB * FLAT));
  CHK(dmalloc(&f.part, part));
  CHK(dmalloc(&f.d1, (size_t)maxB * HID));
  CHK(dmalloc(&f.z, (size_t)maxB * A));
  CHK(dmalloc(&f.p, (size_t)maxB * A));
  CHK(dmalloc(&f.v, (size_t)maxB));
  HIPCHK(hipMalloc((void**)&f.xu8, (size_t)maxB * XS));
  return GA3C_OK;
}

void free_fwd(Fwd& f) {
  for (float* p : {f.x, f.n1, f.n2, f.part, f.d1, f.z, f.p, f.v})
    if (p) (void)hipFree(p);
  if (f.xu8) (void)hipFree(f.xu8);
}

bool is_pinned(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// dense1 forward: the LDS-tiled kernel while its grid is one round of workgroups (one workgroup per CU: its LDS image
// is 120-145 KB), the register-fragment kernel beyond
int launch_dense1_fwd(ga3c_net* net, const float* flat, const float* pk, float* part, int B, int ks, hipStream_t st,
                      hipEvent_t e0, hipEvent_t e1, bool prefer_frag = false) {
  const int max_steps = (KSTEPS_DENSE + ks - 1) / ks;
  const int mt = B <= 128 ? 1 : 2;
  const size_t lds = (size_t)d1f_lds_floats(mt, max_steps) * sizeof(float);
  const int tile_blocks = dense1_fwd_blocks(B, ks, mt);
  if (net->d1f_tile && !prefer_frag && max_steps <= 16 && lds <= 160 * 1024 && tile_blocks <= 256) {
    if (mt == 1)
      hipExtLaunchKernelGGL(dense1_fwd_tile_kernel<1>, dim3(tile_blocks), dim3(512), lds, st, e0, e1, 0, flat, pk, part, B, ks, max_steps);
    else
      hipExtLaunchKernelGGL(dense1_fwd_tile_kernel<2>, dim3(tile_blocks), dim3(512), lds, st, e0, e1, 0, flat, pk, part, B, ks, max_steps);
  } else if (B <= 256) {
    hipExtLaunchKernelGGL(dense1_fwd_kernel<1>, dim3(dense1_fwd_blocks(B, ks, 1)), dim3(256), 0, st, e0, e1, 0, flat, pk, part, B, ks, -1);
  } else {
    hipExtLaunchKernelGGL(dense1_fwd_kernel<2>, dim3(dense1_fwd_blocks(B, ks, 2)), dim3(256), 0, st, e0, e1, 0, flat, pk, part, B, ks, -1);
  }
  HIPCHK(hipGetLastError());
  return GA3C_OK;
}


constexpr int HEADS_WAVES = 1;   // samples (waves) per heads workgroup
// ---- kernel launch helpers (shape checks live here: every grid is derived from B on the host)
// stop_ev: recorded behind the step -- as the completion event of its LAST launch (hipExtLaunchKernelGGL), which saves the
// host the hipEventRecord call and the queue a packet of its own
int launch_forward(ga3c_net* net, const Fwd& f, int idx, int B, hipStream_t st, bool train,
                   const TrainLane* tl, float beta, float* out_p = nullptr, float* out_v = nullptr, hipEvent_t stop_ev = nullptr) {
  if (B < 1 || B > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", B, net->maxB);
  const int A = net->A;
  const float* th = net->theta[idx];
  const void* xin = f.src_off ? (const void*)f.src_base : (f.x_u8 ? (const void*)f.xu8 : (const void*)f.x);
  if (f.src_off && !(net->fused_conv && B <= FUSED_CONV_MAX_B)) return fail(GA3C_ESTATE, "scattered intake needs the fused conv stack");
  if (net->fused_conv && B <= FUSED_CONV_MAX_B) {   // one workgroup per CU: pays off only while a batch is a single wave of workgroups
    const size_t lds = CS_LDS_FLOATS * sizeof(float);
    SrcOffsets so;
    so.n = 0;
    so.n_dst = 0;
    if (f.src_off && f.src_off_host && !net->graphs && net->offsets_in_args) {   // plain launch: the offsets ride in the kernel arguments
      memcpy(so.off, f.src_off_host, (size_t)B * sizeof(int64_t));
      so.n = B;
    }
    if (f.cache_dst && f.src_off && f.x_u8 && !train && !net->graphs) {           // the state cache takes a copy of what is staged
      memcpy(so.dst, f.cache_dst, (size_t)B * sizeof(int64_t));
      so.n_dst = B;
    }
#define CSTACK(T, U)                                                                                                \
  hipLaunchKernelGGL((conv_stack_fwd_kernel<T, U>), dim3(B * 2), dim3(1024), lds, st, xin, net->theta_pk[idx] + PK_W1F, th + OFF_B1, \
                     net->theta_pk[idx] + PK_W2F, th + OFF_B2, f.n1, f.n2, B, f.src_off, so, net->cache_ring)
    if (train) { if (f.x_u8) CSTACK(true, true); else CSTACK(true, false); }
    else { if (f.x_u8) CSTACK(false, true); else CSTACK(false, false); }
#undef CSTACK
  } else {
    if (f.x_u8)
      hipLaunchKernelGGL(conv1_fwd_kernel<true>, dim3(B * 7), dim3(256), 0, st, xin, th + OFF_W1, th + OFF_B1, f.n1, B);
    else
      hipLaunchKernelGGL(conv1_fwd_kernel<false>, dim3(B * 7), dim3(256), 0, st, xin, th + OFF_W1, th + OFF_B1, f.n1, B);
    if (net->c2f_quarter && B > 128 && B <= 192) hipLaunchKernelGGL(conv2_fwd_kernel<true>, dim3(B * 4), dim3(256), 0, st, f.n1, th + OFF_W2, th + OFF_B2, f.n2, B);
    else hipLaunchKernelGGL(conv2_fwd_kernel<false>, dim3(B * 2), dim3(256), 0, st, f.n1, th + OFF_W2, th + OFF_B2, f.n2, B);
  }
  const int ks = dense_ks(B);
  HeadArgs h;
  memset(&h, 0, sizeof h);
  h.part = f.part; h.ks = ks; h.B = B; h.A = A;
  h.bd = th + OFF_BD; h.wv = th + OFF_WV; h.bv = th + OFF_BV; h.wp = th + OFF_WP; h.bp = th + off_bp(A);
  h.d1 = f.d1; h.z = f.z; h.p = out_p ? out_p : f.p; h.v = out_v ? out_v : f.v;   // out_*: pinned host memory, written in place
  h.log_eps = net->cfg.log_epsilon; h.min_policy = net->cfg.min_policy;
  h.log_softmax = (net->cfg.flags & GA3C_FLAG_LOG_SOFTMAX) ? 1 : 0;
  if (train) { h.y_r = tl->yr; h.act = tl->act; h.dz = tl->dz; h.dv = tl->dv; h.lossrow = tl->lossrow; h.dd1 = tl->dd1; h.beta = beta; }
  // several prediction lanes at work: the fragment kernel (no LDS) runs beside the other lanes' conv stacks
  const bool frag = !train && !net->graphs && net->d1f_frag_lanes > 0 && B >= 64 &&
                    net->predict_inflight.load(std::memory_order_relaxed) >= net->d1f_frag_lanes;
  CHK(launch_dense1_fwd(net, f.n2, net->theta_pk[idx], f.part, B, ks, st, nullptr, nullptr, frag));
#define HEADS(T, AM)                                                                                                        \
  do {                                                                                                                      \
    if (stop_ev) hipExtLaunchKernelGGL((heads_kernel<T, AM>), dim3((B + HEADS_WAVES - 1) / HEADS_WAVES), dim3(64 * HEADS_WAVES), 0, st, nullptr, stop_ev, 0, h); \
    else hipLaunchKernelGGL((heads_kernel<T, AM>), dim3((B + HEADS_WAVES - 1) / HEADS_WAVES), dim3(64 * HEADS_WAVES), 0, st, h);                                 \
  } while (0)
  if (A <= 8) { if (train) HEADS(true, 8); else HEADS(false, 8); }
  else if (A <= 24) { if (train) HEADS(true, 24); else HEADS(false, 24); }
  else { if (train) HEADS(true, 64); else HEADS(false, 64); }
#undef HEADS
  HIPCHK(hipGetLastError());
  return GA3C_OK;
}

// conv1_dw workgroups (= partial slabs) of the split conv backward for a batch of B rows (see launch_backward)
int conv1_dw_blocks(const ga3c_net* net, int B) {
  const int units = B * 7;
  int n = 512;
  if (net->c1dw_blocks > 0) n = net->c1dw_blocks;
  else if (B <= 256) {
    const int one_round = 768 - 4 * B;
    n = 5 * one_round >= units ? one_round : 256;
  }
  return units < n ? units : n;
}

// overlap: the caller will apply the gradients right away (train, not compute_grads): start their all-reduce as soon as
// each part of the arena is final; returns with the train stream already waiting for the exchange
// keep_dn1: also store dn1 (consumed on chip by the fused conv backward; kept in HBM for ga3c_net_fetch after compute_grads)
int launch_backward(ga3c_net* net, TrainLane& t, int idx, int B, bool overlap = false, const FusedUpd* fu = nullptr, bool keep_dn1 = true) {
  const int A = net->A;
  const float* th = net->theta[idx];
  FusedUpd upd;
  memset(&upd, 0, sizeof upd);
  if (fu) upd = *fu;
  hipStream_t st = t.st;
  float* g = t.grad;
  HeadBwdArgs hb;
  hb.B = B; hb.A = A; hb.d1 = t.f.d1; hb.dz = t.dz; hb.dv = t.dv; hb.lossrow = t.lossrow;
  hb.g_wp = g + OFF_WP; hb.g_bp = g + off_bp(A); hb.g_wv = g + OFF_WV; hb.g_bv = g + OFF_BV; hb.losses = t.losses;
  // dense1/w stepped inside conv_bwd, beside its MFMA phases, instead of in dense1_bwd_tile's epilogue: worth 0.5 us of the
  // 128-row step (the step's 24 MB cost conv_bwd 2.2 us where they cost the epilogue 3.5) while every workgroup of conv_bwd
  // steps ONE 16-row group; below 121 rows the groups left over are a tail and it loses 0.2 us (profiles/README.md)
  const bool fused_cb = net->conv_bwd_fused && B <= 128 && B >= net->conv_bwd_min;
  upd.defer_wd = upd.on && fused_cb && B <= net->d1b_tile_max &&
                 (net->wd_step_in_conv_bwd >= 2 || (net->wd_step_in_conv_bwd == 1 && 2 * B >= KSTEPS_DENSE));
  // beyond the fused conv_bwd the step rides in conv2_dx (conv2_dx_wd_kernel)
  if (upd.on && !fused_cb && B <= net->d1b_tile_max && net->wd_step_in_conv2_dx) upd.defer_wd = 1;
  if (B <= net->d1b_tile_max) {
    Dense1TileArgs d;
    d.n2 = t.f.n2; d.dd1 = t.dd1; d.wd = th + OFF_WD; d.g_wd = g + OFF_WD; d.g_bd = g + OFF_BD; d.dn2 = t.dn2; d.B = B;
    d.hb = hb;
    d.upd = upd;
    d.role_blocks = A + 2 < 14 ? A + 2 : 14;       // 242 tiles + the roles stay within one round of workgroups on 256 CUs
    d.tail_lds = net->d1b_tail && B > D1B_ROWS && B <= D1B_ROWS + D1B_TAIL_ROWS;
    const size_t d1b_lds = (d.tail_lds ? D1B_LDS_FLOATS_TAIL : D1B_LDS_FLOATS) * sizeof(float);
    if (upd.on && upd.defer_wd && d.tail_lds) hipLaunchKernelGGL((dense1_bwd_tile_kernel<2, true>), dim3(D1B_TILES + d.role_blocks), dim3(1024), d1b_lds, st, d);
    else if (upd.on && upd.defer_wd) hipLaunchKernelGGL(dense1_bwd_tile_kernel<2>, dim3(D1B_TILES + d.role_blocks), dim3(1024), d1b_lds, st, d);
    else if (upd.on && d.tail_lds) hipLaunchKernelGGL((dense1_bwd_tile_kernel<1, true>), dim3(D1B_TILES + d.role_blocks), dim3(1024), d1b_lds, st, d);
    else if (upd.on) hipLaunchKernelGGL(dense1_bwd_tile_kernel<1>, dim3(D1B_TILES + d.role_blocks), dim3(1024), d1b_lds, st, d);
    else if (d.tail_lds) hipLaunchKernelGGL((dense1_bwd_tile_kernel<0, true>), dim3(D1B_TILES + d.role_blocks), dim3(1024), d1b_lds, st, d);
    else hipLaunchKernelGGL(dense1_bwd_tile_kernel<0>, dim3(D1B_TILES + d.role_blocks), dim3(1024), d1b_lds, st, d);
  } else {
    Dense1BwdArgs d;
    d.n2 = t.f.n2; d.dd1 = t.dd1; d.wd = th + OFF_WD; d.g_wd = g + OFF_WD; d.g_bd = g + OFF_BD; d.dn2 = t.dn2; d.B = B;
    d.hb = hb; d.dw_gx = FLAT / 32 + A + 2; d.dw_blocks = 2 * d.dw_gx; d.dx_gx = FLAT / 32;
    d.dx_mt = dense_dx_mt(B);
    const int dx_blocks = d.dx_gx * (((B + 16 * d.dx_mt - 1) / (16 * d.dx_mt) + 3) / 4);
    hipLaunchKernelGGL(dense1_bwd_kernel, dim3(d.dw_blocks + dx_blocks), dim3(256), 0, st, d);
  }
  if (overlap) {
    HIPCHK(hipEventRecord(net->ev_tail_ready, st));
    HIPCHK(hipStreamWaitEvent(net->cst, net->ev_tail_ready, 0));
    NCCLCHK(ncclAllReduce(g + OFF_WD, g + OFF_WD, (size_t)(net->n - OFF_WD), ncclFloat, ncclSum, net->comm, net->cst));
    if (net->head_on_train_stream) HIPCHK(hipEventRecord(net->ev_comm_done, net->cst));   // (waited for below, behind the head's exchange)
  }
  int nch1, nch2;
  if (fused_cb) {   // one workgroup per CU: beyond one round the tail of the second costs more than the fusion saves
    // conv2_dw + conv2_dx + conv1_dw of a sample half in ONE workgroup (conv_bwd_kernel): dn1 never leaves the chip between them
    const int grid = 2 * B;                         // (sample, half); one slab pair per workgroup
    const size_t lds = CB_LDS_FLOATS * sizeof(float);
#define CONV_BWD(U, W, XP)                                                                                                  \
  hipLaunchKernelGGL((conv_bwd_kernel<U, W>), dim3(grid), dim3(1024), lds, st, (const void*)(XP), t.f.n1, t.dn2,              \
                     net->theta_pk[idx] + PK_W2DX, keep_dn1 ? t.dn1 : nullptr, t.slab2, t.slab1, B, g + OFF_WD, upd)
    const bool wd = upd.on && upd.defer_wd;
    if (t.f.x_u8) { if (wd) CONV_BWD(true, true, t.f.xu8); else CONV_BWD(true, false, t.f.xu8); }
    else { if (wd) CONV_BWD(false, true, t.f.x); else CONV_BWD(false, false, t.f.x); }
#undef CONV_BWD
    nch1 = nch2 = grid;
  } else {
    nch2 = B < 256 ? B : 256;          // sample groups = partial slabs
    // workgroups = partial slabs.  Measured at batch 128 (round 2): 512 / 384 / 256 / 192 / 128 workgroups -> train step
    // 68.8 / 68.7 / 68.0 / 69.8 / 73.0 us: 256 (3.5 units each, 4.2 MB of slabs instead of 8.4) is as fast
    // ... but uint8 states (4-byte loads per pixel, a longer chain per unit) lose 4 % at 256 (13.6 k vs 14.2 k steps/s), and the
    // two input formats must cut the units alike to stay bit-identical: 512 for both
    // Round 4: behind conv2_dx, conv1_dw shares ONE launch with conv2_dw (conv_dw_pair_kernel: three workgroups of either kind
    // per CU = 768 slots).  With 768 - 4 B workgroups -- 240 at 132 rows, 3.9 units each -- the pair is one round and the slab
    // reduction reads half as much: 132-row step 70.2 -> 66.0 us (f32), 70.0 -> 67.1 (uint8, whose staging became 16-byte loads
    // for it: with dword loads it LOST 1.6 us at 240); 200 workgroups: 67.2.  Past ~142 rows that count would mean more than five
    // units per workgroup: 256 (150 / 160 rows: 74.9 / 75.9 us against 77.0 / 77.9 at 512).  A function of B alone, so that
    // every form of the step (paired or not, uint8 or f32) cuts the slabs alike and gives the same bits.
    nch1 = conv1_dw_blocks(net, B);
    const bool pair = net->dw_pair && B <= 256;    // conv2_dw + conv1_dw side by side in one launch, behind conv2_dx
    // conv2's two gradients are separate launches: their LDS/VGPR budgets differ too much to share one grid
    if (!pair) {
      if (net->c2dw_occ >= 3 && B <= 256) hipLaunchKernelGGL(conv2_dw_kernel<3>, dim3(nch2, 4), dim3(256), 0, st, t.f.n1, t.dn2, t.slab2, B);
      else hipLaunchKernelGGL(conv2_dw_kernel<2>, dim3(nch2, 4), dim3(256), 0, st, t.f.n1, t.dn2, t.slab2, B);
    }
    if (upd.on && upd.defer_wd)
      hipLaunchKernelGGL(conv2_dx_wd_kernel, dim3(B + C2DX_WD_BLOCKS, 2), dim3(512), 0, st, t.dn2, net->theta_pk[idx] + PK_W2DX, t.f.n1, t.dn1, B,
                         (const float*)(g + OFF_WD), upd, net->wd_blocks_first);
    else
      hipLaunchKernelGGL(conv2_dx_kernel, dim3(B, 2), dim3(512), 0, st, t.dn2, net->theta_pk[idx] + PK_W2DX, t.f.n1, t.dn1, B);
    if (pair) {
      if (t.f.x_u8)
        hipLaunchKernelGGL(conv_dw_pair_kernel<true>, dim3(nch1 + 4 * nch2), dim3(256), 0, st, (const void*)t.f.xu8, t.dn1, t.slab1, B * 7, nch1,
                           t.f.n1, t.dn2, t.slab2, B, nch2);
      else
        hipLaunchKernelGGL(conv_dw_pair_kernel<false>, dim3(nch1 + 4 * nch2), dim3(256), 0, st, (const void*)t.f.x, t.dn1, t.slab1, B * 7, nch1,
                           t.f.n1, t.dn2, t.slab2, B, nch2);
    } else if (t.f.x_u8)
      hipLaunchKernelGGL(conv1_dw_kernel<true>, dim3(nch1), dim3(256), 0, st, (const void*)t.f.xu8, t.dn1, t.slab1, B * 7);
    else
      hipLaunchKernelGGL(conv1_dw_kernel<false>, dim3(nch1), dim3(256), 0, st, (const void*)t.f.x, t.dn1, t.slab1, B * 7);
  }
  {
    SlabSet s1{t.slab1, nch1, SLAB1, 256 * 16, g + OFF_W1, g + OFF_B1, (SLAB1 + 63) / 64, OFF_W1, OFF_B1};
    SlabSet s2{t.slab2, nch2, SLAB2, 256 * 32, g + OFF_W2, g + OFF_B2, (SLAB2 + 63) / 64, OFF_W2, OFF_B2};
    if (upd.on) hipLaunchKernelGGL(slab_reduce_kernel<true>, dim3(s1.nblocks + s2.nblocks), dim3(1024), 0, st, s1, s2, upd);
    else hipLaunchKernelGGL(slab_reduce_kernel<false>, dim3(s1.nblocks + s2.nblocks), dim3(1024), 0, st, s1, s2, upd);
  }
  HIPCHK(hipGetLastError());
  if (overlap && net->head_on_train_stream) {
    // The conv gradients (12 k floats) are complete only now, with nothing left to overlap their exchange with: it goes on the
    // TRAIN stream itself, in line, instead of a hop to the comm stream and back (two cross-queue dependencies, ~7 us each on
    // the GPU, and two more API calls on a host that is the bottleneck of this path: profiles/r04_dp_1rank_timeline.txt).
    // RCCL runs the collectives of one communicator in the order they were issued -- identical on every rank --, so this one
    // starts behind the big one on the comm stream; the optimizer step then waits for both.
    NCCLCHK(ncclAllReduce(g, g, (size_t)OFF_WD, ncclFloat, ncclSum, net->comm, st));
    HIPCHK(hipStreamWaitEvent(st, net->ev_comm_done, 0));
  } else if (overlap) {
    HIPCHK(hipEventRecord(net->ev_head_ready, st));
    HIPCHK(hipStreamWaitEvent(net->cst, net->ev_head_ready, 0));
    NCCLCHK(ncclAllReduce(g, g, (size_t)OFF_WD, ncclFloat, ncclSum, net->comm, net->cst));
    HIPCHK(hipEventRecord(net->ev_comm_done, net->cst));
    HIPCHK(hipStreamWaitEvent(st, net->ev_comm_done, 0));
  }
  return GA3C_OK;
}

int launch_rmsprop(ga3c_net* net, const float* grad, float* scales, const float* tin, float* tout, float* pk_out,
                   float lr, hipStream_t st) {
  const bool clip = net->cfg.flags & GA3C_FLAG_GRAD_CLIP;
  const bool mom = net->cfg.rmsprop_momentum != 0.0f;
  const float omr = 1.0f - net->cfg.rmsprop_decay;
  const int blocks = RMS_WD_BLOCKS + (int)((net->n - (int64_t)FLAT * HID + 255) / 256);
  if (clip)
    hipLaunchKernelGGL(clip_scale_kernel, dim3(10), dim3(256), 0, st, grad, net->tt, net->cfg.grad_clip_norm, scales);
#define RMS(C, M)                                                                                                \
  hipLaunchKernelGGL((rmsprop_kernel<C, M>), dim3(blocks), dim3(256), 0, st, tin, tout, net->ms, net->mom,        \
                     grad, net->n, lr, omr, net->cfg.rmsprop_momentum, net->cfg.rmsprop_epsilon, net->tt, scales,  \
                     pk_out)
  if (clip && mom) RMS(true, true);
  else if (clip) RMS(true, false);
  else if (mom) RMS(false, true);
  else RMS(false, false);
#undef RMS
  HIPCHK(hipGetLastError());
  return GA3C_OK;
}

// forward on a prediction lane: pick the current weights under the shared lock
// the kernels of one prediction step on lane L: the intake gather (offsets already in L.h_off), then the forward pass
int launch_step(ga3c_net* net, Lane& L, int idx, int B, int mode, float* out_p, float* out_v, hipEvent_t stop_ev = nullptr) {
  L.f.src_base = nullptr;
  L.f.src_off = nullptr;
  if (mode != STEP_RESIDENT) {
    const bool u8 = mode != STEP_GATHER_F32;
    const uint8_t* base = mode == STEP_QUEUES ? reinterpret_cast<const uint8_t*>(net->fr.stacks) : net->reg_dev;
    L.f.x_u8 = u8;
    if (net->fused_conv && B <= FUSED_CONV_MAX_B) {
      // small batches (every engine batch): the conv stack reads the scattered states itself -- one launch less
      L.f.src_base = base;
      L.f.src_off = L.h_off;
      L.f.src_off_host = L.h_off;
      L.f.cache_dst = (L.cache_on && mode == STEP_GATHER_U8) ? L.cache_dst : nullptr;
    } else {
      const SmallCopy none{nullptr, nullptr, 0, nullptr, nullptr, 0};
      RowOffsets ro;
      ro.n = 0;                                              // (beyond the fused conv stack's range: batches of more than 128 rows)
      if (u8) hipLaunchKernelGGL(gather_rows_kernel<XS / 16>, dim3(gather_blocks(B, XS / 16, net->gather_max_blocks)), dim3(256), 0, L.st, base, L.h_off, reinterpret_cast<uint4*>(L.f.xu8), B, none, ro);
      else hipLaunchKernelGGL(gather_rows_kernel<XS / 4>, dim3(gather_blocks(B, XS / 4, net->gather_max_blocks)), dim3(256), 0, L.st, base, L.h_off, reinterpret_cast<uint4*>(L.f.x), B, none, ro);
      if (L.cache_on && mode == STEP_GATHER_U8)               // ... and the state cache takes its copy from the gathered batch
        hipLaunchKernelGGL(file_rows_kernel<XS / 16>, dim3((XS / 16 + 255) / 256, B), dim3(256), 0, L.st, reinterpret_cast<const uint4*>(L.f.xu8),
                           net->cache_ring, L.cache_dst, B);
    }
  }
  const int rc = launch_forward(net, L.f, idx, B, L.st, false, nullptr, 0.f, out_p, out_v, stop_ev);
  L.f.src_base = nullptr;
  L.f.src_off = nullptr;
  L.f.src_off_host = nullptr;
  L.f.cache_dst = nullptr;
  return rc;
}

void drop_graphs(Lane& L) {
  for (auto& kv : L.graphs) (void)hipGraphExecDestroy(kv.second);
  L.graphs.clear();
}

// launch_step, replayed from the lane's graph cache (captured on first use of a shape)
// mark_done: L.done is recorded behind the step (the callers that will wait for it, or let a train step wait for it)
int lane_step(ga3c_net* net, Lane& L, int idx, int B, int mode, float* out_p, float* out_v, bool mark_done = false) {
  if (!net->graphs) {
    if (mark_done && net->stop_events) return launch_step(net, L, idx, B, mode, out_p, out_v, L.done);
    CHK(launch_step(net, L, idx, B, mode, out_p, out_v));
    if (mark_done) HIPCHK(hipEventRecord(L.done, L.st));
    return GA3C_OK;
  }
  const bool u8 = mode == STEP_RESIDENT ? L.f.x_u8 : mode != STEP_GATHER_F32;
  const int64_t key = ((int64_t)B << 8) | (idx << 5) | (mode << 2) | (u8 ? 2 : 0) | (out_p ? 1 : 0);
  auto it = L.graphs.find(key);
  if (it == L.graphs.end()) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    HIPCHK(hipStreamBeginCapture(L.st, hipStreamCaptureModeThreadLocal));
    const int rc = launch_step(net, L, idx, B, mode, out_p, out_v);
    const hipError_t e = hipStreamEndCapture(L.st, &g);
    if (rc != GA3C_OK) {
      if (g) (void)hipGraphDestroy(g);
      return rc;
    }
    if (e != hipSuccess) return fail(GA3C_EHIP, "stream capture of a prediction step failed: %s", hipGetErrorString(e));
    const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) return fail(GA3C_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei));
    it = L.graphs.emplace(key, ex).first;
  }
  L.f.x_u8 = u8;
  HIPCHK(hipGraphLaunch(it->second, L.st));
  if (mark_done) HIPCHK(hipEventRecord(L.done, L.st));
  return GA3C_OK;
}

// cur <- the newest buffer whose step has finished (ready_mu held; wmu held shared or unique).  An event the optimizer did
// not record (no predictions were around) is recorded now, at the tail of the train stream: it then completes no earlier
// than the step it stands for.
int adopt_finished(ga3c_net* net) {
  const int c = net->cur.load(), l = net->latest;
  if (l == c) return GA3C_OK;
  const int m = 3 - l - c;                                   // the buffer enqueued between them, if any
  for (int cand : {l, m}) {
    if (net->wseq[cand] <= net->wseq[c]) continue;
    if (!net->event_valid[cand]) {
      HIPCHK(hipEventRecord(net->theta_ready[cand], net->tr.st));
      net->event_valid[cand] = true;
    }
    if (hipEventQuery(net->theta_ready[cand]) == hipSuccess) {
      net->cur.store(cand);
      return GA3C_OK;
    }
  }
  return GA3C_OK;
}

int lane_forward(ga3c_net* net, Lane& L, int B, int mode, float* out_p, float* out_v) {
  std::shared_lock<std::shared_mutex> lk(net->wmu);
  net->pred_seq.fetch_add(1);
  if (net->latest != net->cur.load()) {
    std::lock_guard<std::mutex> g(net->ready_mu);
    CHK(adopt_finished(net));
  }
  const int idx = net->cur.load();      // a finished step's weights: nothing to wait for
  if (net->time_predictions) HIPCHK(hipEventRecord(L.tm0, L.st));
  CHK(lane_step(net, L, idx, B, mode, out_p, out_v, true));   // L.done: behind the step
  if (net->time_predictions) HIPCHK(hipEventRecord(L.tm1, L.st));
  L.dirty[idx] = true;
  return GA3C_OK;
}

// The buffer the next optimizer step writes: neither `latest` (the step reads it) nor `cur` (predictions read it).  The
// train stream waits for every prediction lane that has read that buffer since it was last written.  Cross-stream events
// cost several microseconds of queue idle each on this stack, so they are used only when a lane has really touched it.
int claim_other_buffer(ga3c_net* net, TrainLane& t, int* idx_out, int* other_out) {
  std::unique_lock<std::shared_mutex> lk(net->wmu);
  const int idx = net->latest, c = net->cur.load();
  const int other = c == idx ? (idx + 1) % 3 : 3 - idx - c;
  for (Lane* L : net->lanes) {
    if (L->dirty[other]) {
      // the lane's last step is at least as late as its last read of `other`: if it has finished (the usual case -- the
      // buffer was `cur` two steps ago) there is nothing to order; a wait costs the host a call and the train stream a
      // barrier packet (several microseconds of queue idle each on this stack)
      const hipError_t q = hipEventQuery(L->done);
      if (q != hipSuccess) {
        (void)hipGetLastError();                           // hipErrorNotReady is not an error here
        HIPCHK(hipStreamWaitEvent(t.st, L->done, 0));
        stat_add(net, GA3C_STAT_TRAIN_READER_WAITS, 1);
      }
      L->dirty[other] = false;
    }
  }
  // From here until the step is published the buffer holds no weights anyone may adopt: it may be the in-between buffer of
  // steps enqueued back to back (newer than `cur`, its old step long finished), which adopt_finished would otherwise hand to
  // a prediction that this step, already enqueued, does not wait for.
  net->wseq[other] = 0;
  net->event_valid[other] = false;
  *idx_out = idx;
  *other_out = other;
  return GA3C_OK;
}

// the optimizer step that wrote theta[other] has been enqueued on the train stream
int publish_other_buffer(ga3c_net* net, TrainLane& t, int other) {
  std::unique_lock<std::shared_mutex> lk(net->wmu);
  const uint64_t seq = net->pred_seq.load();
  net->event_valid[other] = seq != net->pred_seen;   // predictions are arriving: give them an event to poll
  net->pred_seen = seq;
  if (net->event_valid[other]) HIPCHK(hipEventRecord(net->theta_ready[other], t.st));
  net->latest = other;
  net->wseq[other] = ++net->wcount;
  net->step.fetch_add(1);
  t.wrote = other;
  t.wrote_seq = net->wseq[other];
  return GA3C_OK;
}

// the step number `seq` that wrote theta[buf] is known to have finished (its caller has waited for it): predictions may read
// it -- unless a later step has claimed the buffer since (wseq differs: that step's own waiter will hand it over)
void adopt_buffer(ga3c_net* net, int buf, uint64_t seq) {
  if (buf < 0) return;
  std::shared_lock<std::shared_mutex> lk(net->wmu);
  std::lock_guard<std::mutex> g(net->ready_mu);
  if (net->wseq[buf] == seq && seq > net->wseq[net->cur.load()]) net->cur.store(buf);
}

// gradients of the batch staged in train lane `t` -> t.grad (the lane's mutex is held by the caller).  will_apply: the
// caller steps the optimizer right away (train, not compute_grads).  Then (a) with a communicator the exchange is started
// inside the backward pass, and (b) without one, and without clipping, the backward kernels apply RMSProp themselves to
// the elements whose gradient they complete (FusedUpd): train_apply then has nothing left to launch.
int train_grads(ga3c_net* net, TrainLane& t, int B, float beta, bool will_apply = false, float lr = 0.f) {
  const bool fuse = will_apply && net->fused_update && !net->comm && !(net->cfg.flags & GA3C_FLAG_GRAD_CLIP) &&
                    B <= net->d1b_tile_max;
  int idx, other;
  if (fuse && !net->hogwild) {
    CHK(claim_other_buffer(net, t, &idx, &other));
  } else {
    std::shared_lock<std::shared_mutex> lk(net->wmu);
    idx = other = net->latest;   // the newest weights: written by this same stream (synchronous mode) or in place (Hogwild)
  }
  CHK(launch_forward(net, t.f, idx, B, t.st, true, &t, beta));
  t.exchanged = will_apply && net->comm && net->comm_overlap && !net->hogwild && &t == &net->tr;
  FusedUpd fu;
  memset(&fu, 0, sizeof fu);
  if (fuse) {
    fu.tin = net->theta[idx]; fu.tout = net->theta[other]; fu.ms = net->ms; fu.mom = net->mom; fu.pk = net->theta_pk[other];
    fu.lr = lr; fu.omr = 1.0f - net->cfg.rmsprop_decay; fu.mu = net->cfg.rmsprop_momentum; fu.eps = net->cfg.rmsprop_epsilon;
    fu.on = 1;
  }
  CHK(launch_backward(net, t, idx, B, t.exchanged, fuse ? &fu : nullptr, !will_apply));
  t.stepped = fuse;
  t.stepped_other = other;
  return GA3C_OK;
}

int train_apply(ga3c_net* net, TrainLane& t, float lr) {
  if (t.stepped) {           // the backward kernels have applied the step (train_grads, FusedUpd)
    t.stepped = false;
    if (net->hogwild) {
      net->step.fetch_add(1);
      return GA3C_OK;
    }
    return publish_other_buffer(net, t, t.stepped_other);
  }
  if (net->hogwild) {
    // The reference's NT trainer threads run sess.run(train_op) concurrently on shared variables without locking
    // (Server.py:132-134, TF use_locking=False): every lane updates theta / ms in place from its own stream; reads by
    // other lanes may see a step half applied, and two optimizer kernels may race on an element.
    const int idx = net->latest;
    CHK(launch_rmsprop(net, t.grad, t.scales, net->theta[idx], net->theta[idx], net->theta_pk[idx], lr, t.st));
    net->step.fetch_add(1);
    return GA3C_OK;
  }
  if (net->comm && !t.exchanged)       // (train_grads has already exchanged the gradients when it knew a step would follow)
    NCCLCHK(ncclAllReduce(t.grad, t.grad, (size_t)net->n, ncclFloat, ncclSum, net->comm, t.st));
  t.exchanged = false;
  int idx, other;
  CHK(claim_other_buffer(net, t, &idx, &other));
  CHK(launch_rmsprop(net, t.grad, t.scales, net->theta[idx], net->theta[other], net->theta_pk[other], lr, t.st));
  return publish_other_buffer(net, t, other);
}

// the lane's currently bound input buffers, staged on the train stream itself (resident / evaluation paths)
Stage lane_stage(TrainLane& t) { return Stage{t.f.x, t.f.xu8, t.yr, t.act, t.h_in, t.h_off, t.st, t.f.x_u8}; }
Stage intake_stage(TrainLane& t, Intake& in) { return Stage{in.x, in.xu8, in.yr, in.act, in.h_in, in.h_off, t.gst, in.x_u8}; }

// make `in` the batch the lane's next step reads (the lane's mutex is held; `in` is held by the same thread)
void bind_intake(TrainLane& t, Intake& in) {
  t.f.x = in.x; t.f.xu8 = in.xu8; t.f.x_u8 = in.x_u8;
  t.yr = in.yr; t.act = in.act; t.h_in = in.h_in; t.h_off = in.h_off;
  t.losses = in.losses;
}

int stage_train_inputs(ga3c_net* net, Stage& s, const void* x, bool u8, const float* y_r, const float* a, int B) {
  if (B < 1 || B > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", B, net->maxB);
  if (x) {
    const size_t xb = (size_t)B * XS * (u8 ? 1 : sizeof(float));
    void* dst = u8 ? (void*)s.xu8 : (void*)s.x;
    if (is_pinned(x)) {
      HIPCHK(hipMemcpyAsync(dst, x, xb, hipMemcpyHostToDevice, s.st));
    } else {
      memcpy(s.h_in, x, xb);
      HIPCHK(hipMemcpyAsync(dst, s.h_in, xb, hipMemcpyHostToDevice, s.st));
    }
    s.x_u8 = u8;
  }
  float* hy = s.h_in + (size_t)net->maxB * XS;
  float* ha = hy + net->maxB;
  if (y_r) {
    memcpy(hy, y_r, (size_t)B * sizeof(float));
    HIPCHK(hipMemcpyAsync(s.yr, hy, (size_t)B * sizeof(float), hipMemcpyHostToDevice, s.st));
  }
  if (a) {
    memcpy(ha, a, (size_t)B * net->A * sizeof(float));
    HIPCHK(hipMemcpyAsync(s.act, ha, (size_t)B * net->A * sizeof(float), hipMemcpyHostToDevice, s.st));
  }
  return GA3C_OK;
}

// resident / evaluation callers: stage into the lane's bound buffers on the train stream
int stage_train_inputs(ga3c_net* net, TrainLane& t, const void* x, bool u8, const float* y_r, const float* a, int B) {
  Stage s = lane_stage(t);
  CHK(stage_train_inputs(net, s, x, u8, y_r, a, B));
  t.f.x_u8 = s.x_u8;
  return GA3C_OK;
}

int read_losses(ga3c_net* net, TrainLane& t, float* losses) {
  const float* hl = t.losses;   // pinned host memory, written by the step itself
  HIPCHK(hipStreamSynchronize(t.st));
  if (!net->hogwild) {
    // everything this lane enqueued has finished: predictions read the newest weights from now on (a caller that trains
    // and then predicts sees the new weights)
    std::shared_lock<std::shared_mutex> lk(net->wmu);
    std::lock_guard<std::mutex> g(net->ready_mu);
    net->cur.store(net->latest);
  }
  if (losses) memcpy(losses, hl, 3 * sizeof(float));
  return GA3C_OK;
}

// Network.log (NetworkVP.py:259-265): forward + loss of the batch staged in train lane `t` on the newest weights, no
// backward pass, no update.  losses = {cost_p_1_agg, cost_p_2_agg, cost_v}; d1 / v / p (each may be null) receive the
// activations the reference histograms (NetworkVP_discrate.py:143-146).
int evaluate_staged(ga3c_net* net, TrainLane& t, int B, float beta, float* losses, float* d1, float* v, float* p) {
  int idx;
  {
    std::shared_lock<std::shared_mutex> lk(net->wmu);
    idx = net->latest;
  }
  CHK(launch_forward(net, t.f, idx, B, t.st, true, &t, beta));
  HeadBwdArgs hb;
  memset(&hb, 0, sizeof hb);
  hb.B = B; hb.A = net->A; hb.lossrow = t.lossrow; hb.losses = t.losses;
  hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(256), 0, t.st, hb);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(t.st));
  if (losses) memcpy(losses, t.losses, 3 * sizeof(float));
  if (d1) HIPCHK(hipMemcpy(d1, t.f.d1, (size_t)B * HID * sizeof(float), hipMemcpyDeviceToHost));
  if (v) HIPCHK(hipMemcpy(v, t.f.v, (size_t)B * sizeof(float), hipMemcpyDeviceToHost));
  if (p) HIPCHK(hipMemcpy(p, t.f.p, (size_t)B * net->A * sizeof(float), hipMemcpyDeviceToHost));
  return GA3C_OK;
}

// rows of a batch gathered from the registered host segment into x (device), on stream st
// per-row byte offsets into the registered host segment, validated and copied into the lane's pinned array
int stage_offsets(ga3c_net* net, const int64_t* offsets, int B, bool u8, int64_t* h_off) {
  if (!net->reg_dev) return fail(GA3C_ESTATE, "no host segment registered (ga3c_net_register_host)");
  if (B < 1 || B > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", B, net->maxB);
  const int64_t sb = (int64_t)XS * (u8 ? 1 : 4);
  for (int i = 0; i < B; ++i) {
    if (offsets[i] < 0 || offsets[i] + sb > net->reg_bytes || (offsets[i] & 15))
      return fail(GA3C_EINVAL, "row %d: offset %lld outside the registered segment or not 16-byte aligned", i, (long long)offsets[i]);
    h_off[i] = offsets[i];
  }
  return GA3C_OK;
}

// rows of a batch gathered from the registered host segment into x (device), on stream st; the gather kernel reads
// the offsets out of the pinned host array itself: no H2D copy to wait for
// y_r / a (either may be null) ride along: they are put into the pinned staging array and copied by the gather kernel itself
int launch_gather(ga3c_net* net, const int64_t* offsets, int B, bool u8, Stage& s, const float* y_r = nullptr, const float* a = nullptr) {
  CHK(stage_offsets(net, offsets, B, u8, s.h_off));
  float* hy = s.h_in + (size_t)net->maxB * XS;
  float* ha = hy + net->maxB;
  SmallCopy sc{nullptr, nullptr, 0, nullptr, nullptr, 0};
  if (y_r) {
    memcpy(hy, y_r, (size_t)B * sizeof(float));
    sc.src0 = hy; sc.dst0 = s.yr; sc.n0 = B;
  }
  if (a) {
    memcpy(ha, a, (size_t)B * net->A * sizeof(float));
    sc.src1 = ha; sc.dst1 = s.act; sc.n1 = B * net->A;
  }
  RowOffsets ro;
  ro.n = 0;
  if (B <= 192 && net->offsets_in_args) {                    // the engine's train batches: offsets in the kernel arguments
    memcpy(ro.off, s.h_off, (size_t)B * sizeof(int64_t));
    ro.n = B;
  }
  if (u8) {
    hipLaunchKernelGGL(gather_rows_kernel<XS / 16>, dim3(gather_blocks(B, XS / 16, net->gather_max_blocks)), dim3(256), 0, s.st, net->reg_dev, s.h_off, reinterpret_cast<uint4*>(s.xu8), B, sc, ro);
  } else {
    hipLaunchKernelGGL(gather_rows_kernel<XS / 4>, dim3(gather_blocks(B, XS / 4, net->gather_max_blocks)), dim3(256), 0, s.st, net->reg_dev, s.h_off, reinterpret_cast<uint4*>(s.x), B, sc, ro);
  }
  s.x_u8 = u8;
  HIPCHK(hipGetLastError());
  return GA3C_OK;
}

// captured steps hold the addresses of the registered segment / the frame queues: forget them when those move
void drop_all_graphs(ga3c_net* net) {
  for (Lane* L : net->lanes) {
    std::lock_guard<std::mutex> g(L->mu);
    drop_graphs(*L);
  }
}

void free_frames(Frames& f) {
  if (f.d_tab) (void)hipFree(f.d_tab);
  if (f.stacks) (void)hipFree(f.stacks);
  if (f.d_rgb) (void)hipFree(f.d_rgb);
  if (f.d_planes) (void)hipFree(f.d_planes);
  if (f.h_rgb) (void)hipHostFree(f.h_rgb);
  if (f.h_planes) (void)hipHostFree(f.h_planes);
  if (f.h_agents) (void)hipHostFree(f.h_agents);
  if (f.h_reset) (void)hipHostFree(f.h_reset);
  if (f.h_src) (void)hipHostFree(f.h_src);
  if (f.h_slot) (void)hipHostFree(f.h_slot);
  if (f.ring) (void)hipFree(f.ring);
  f.h_src = nullptr; f.h_slot = nullptr; f.ring = nullptr; f.hist = 0;
  if (f.st) stream_give_back(f.st);
  f.d_tab = nullptr; f.stacks = nullptr; f.d_rgb = f.h_rgb = f.d_planes = f.h_planes = f.h_reset = nullptr;
  f.h_agents = nullptr; f.st = nullptr; f.on = false;
}

// Device-visible address of `bytes` caller bytes at p: inside the registered transport segment, pinned host memory
// and device memory are used where they lie; pageable memory returns nullptr (the caller stages it).
const uint8_t* device_visible(ga3c_net* net, const void* p, size_t bytes) {
  const uint8_t* q = static_cast<const uint8_t*>(p);
  if (net->reg_host) {
    const uint8_t* b = static_cast<const uint8_t*>(net->reg_host);
    if (q >= b && q + bytes <= b + net->reg_bytes) return net->reg_dev + (q - b);
  }
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice) return static_cast<const uint8_t*>(at.devicePointer);
  return nullptr;
}

// one launch of the front-end over n frames; agents/reset/planes may be null (see FrameArgs)
int launch_frames(ga3c_net* net, const uint8_t* rgb_dev, const int32_t* agents, const uint8_t* reset, uint8_t* planes,
                  int n, const int64_t* src_off = nullptr, hipStream_t st = nullptr, const int32_t* slots = nullptr) {
  Frames& f = net->fr;
  if (!st) st = f.st;
  FrameArgs a;
  a.rgb = rgb_dev; a.agents = agents; a.reset = reset; a.planes = planes; a.stacks = f.stacks;
  a.src_off = src_off; a.ring = agents ? f.ring : nullptr; a.ring_slot = slots ? slots : f.h_slot; a.hist = f.hist;
  a.hb = f.hb; a.hk = f.hk; a.vb = f.vb; a.vk = f.vk;
  a.H = f.H; a.W = f.W; a.C = f.C; a.OH = IMG; a.OW = IMG; a.hks = f.hks; a.vks = f.vks;
  if (reinterpret_cast<uintptr_t>(rgb_dev) & 3) return fail(GA3C_EINVAL, "frames: the frame buffer must be 4-byte aligned");
  if (f.C == 1) hipLaunchKernelGGL(plane_push_kernel, dim3(n), dim3(256), 0, st, a);
  else if (f.C == 3) hipLaunchKernelGGL(frame_frontend_kernel<3>, dim3(n), dim3(FE_THREADS), f.lds, st, a);
  else hipLaunchKernelGGL(frame_frontend_kernel<4>, dim3(n), dim3(FE_THREADS), f.lds, st, a);
  HIPCHK(hipGetLastError());
  return GA3C_OK;
}

struct PredictInFlight {   // one per prediction call / lane driver: how many lanes are at work right now
  ga3c_net* n;
  explicit PredictInFlight(ga3c_net* net) : n(net) { n->predict_inflight.fetch_add(1, std::memory_order_relaxed); }
  ~PredictInFlight() { n->predict_inflight.fetch_sub(1, std::memory_order_relaxed); }
  PredictInFlight(const PredictInFlight&) = delete;
  PredictInFlight& operator=(const PredictInFlight&) = delete;
};

// A free lane for one prediction call, locked.  With more lanes than prediction streams, a free lane whose stream no other
// lane is using right now is preferred: two calls in flight then sit on two streams, not behind each other on one.
Lane* take_lane(ga3c_net* net) {
  const unsigned start = net->rr.fetch_add(1);
  const size_t n = net->lanes.size();
  for (int pass = 0; pass < 2; ++pass) {
    for (size_t k = 0; k < n; ++k) {
      Lane* c = net->lanes[(start + k) % n];
      if (pass == 0 && net->stream_busy[c->sidx].load(std::memory_order_relaxed) != 0) continue;
      if (c->mu.try_lock()) {
        net->stream_busy[c->sidx].fetch_add(1, std::memory_order_relaxed);
        c->staged_gen = 0;                                   // the call will put its own rows into the lane's x
        return c;
      }
    }
  }
  Lane* L = net->lanes[start % n];
  const int64_t t0 = now_ns();
  L->mu.lock();
  stat_add(net, GA3C_STAT_PREDICT_LANE_WAIT_NS, now_ns() - t0);
  net->stream_busy[L->sidx].fetch_add(1, std::memory_order_relaxed);
  L->staged_gen = 0;
  return L;
}

// releases what take_lane took (the lane's mutex and its claim on the stream)
struct LaneGuard {
  ga3c_net* net;
  Lane* L;
  LaneGuard(ga3c_net* n, Lane* l) : net(n), L(l) {}
  ~LaneGuard() {
    net->stream_busy[L->sidx].fetch_sub(1, std::memory_order_relaxed);
    L->mu.unlock();
  }
  LaneGuard(const LaneGuard&) = delete;
  LaneGuard& operator=(const LaneGuard&) = delete;
};

// after lane_wait: the GPU span of the step lane_forward enqueued (diagnostic switch)
void note_predict_span(ga3c_net* net, Lane* L) {
  float ms = 0.f;
  if (!net->time_predictions) return;
  (void)hipEventSynchronize(L->tm1);                         // it sits behind the step's own completion event
  if (hipEventElapsedTime(&ms, L->tm0, L->tm1) == hipSuccess) stat_add(net, GA3C_STAT_PREDICT_GPU_NS, (int64_t)(ms * 1e6f));
}

// wait for the step lane_forward enqueued last (it left L->done behind it)
int lane_wait_step(Lane* L) {
  HIPCHK(hipEventSynchronize(L->done));
  return GA3C_OK;
}

// wait until everything lane L has enqueued is done (L's mutex is held)
int lane_wait(Lane* L) {
  if (!L->shared_st) {
    HIPCHK(hipStreamSynchronize(L->st));
    return GA3C_OK;
  }
  HIPCHK(hipEventRecord(L->done, L->st));
  HIPCHK(hipEventSynchronize(L->done));
  return GA3C_OK;
}

int finish_predict(ga3c_net* net, Lane* L, int B, int mode, float* p, float* v, float* z) {
  const int A = net->A;
  float* hp = L->h_out;
  float* hv = hp + (size_t)net->maxB * A;
  float* hz = hv + net->maxB;
  // the heads kernel stores p and v straight into the lane's pinned host buffer: no D2H copies on the round trip
  TraceRange range("ga3c.predict");
  const int64_t t0 = now_ns();
  CHK(lane_forward(net, *L, B, mode, hp, hv));
  if (z) HIPCHK(hipMemcpyAsync(hz, L->f.z, (size_t)B * A * sizeof(float), hipMemcpyDeviceToHost, L->st));
  const int64_t t1 = now_ns();
  if (z) CHK(lane_wait(L));                                  // the copy is behind the step's own event
  else CHK(lane_wait_step(L));
  note_predict_span(net, L);
  stat_add(net, GA3C_STAT_PREDICT_CALLS, 1);
  stat_add(net, GA3C_STAT_PREDICT_ROWS, B);
  stat_add(net, GA3C_STAT_PREDICT_LAUNCH_NS, t1 - t0);
  stat_add(net, GA3C_STAT_PREDICT_SYNC_NS, now_ns() - t1);
  memcpy(p, hp, (size_t)B * A * sizeof(float));
  memcpy(v, hv, (size_t)B * sizeof(float));
  if (z) memcpy(z, hz, (size_t)B * A * sizeof(float));
  return GA3C_OK;
}

int predict_common(ga3c_net* net, const void* x, bool u8, int B, float* p, float* v, float* z) {
  if (!net || !x || !p || !v) return fail(GA3C_EINVAL, "null argument");
  if (B < 1 || B > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", B, net->maxB);
  HIPCHK(hipSetDevice(net->cfg.device));
  PredictInFlight inflight(net);
  Lane* L = take_lane(net);
  LaneGuard guard(net, L);
  if (u8) {
    const size_t nb = (size_t)B * XS;
    if (is_pinned(x)) {
      HIPCHK(hipMemcpyAsync(L->f.xu8, x, nb, hipMemcpyHostToDevice, L->st));
    } else {
      memcpy(L->h_in, x, nb);
      HIPCHK(hipMemcpyAsync(L->f.xu8, L->h_in, nb, hipMemcpyHostToDevice, L->st));
    }
    L->f.x_u8 = true;
  } else {
    const size_t nb = (size_t)B * XS * sizeof(float);
    if (is_pinned(x)) {
      HIPCHK(hipMemcpyAsync(L->f.x, x, nb, hipMemcpyHostToDevice, L->st));
    } else {
      memcpy(L->h_in, x, nb);
      HIPCHK(hipMemcpyAsync(L->f.x, L->h_in, nb, hipMemcpyHostToDevice, L->st));
    }
    L->f.x_u8 = false;
  }
  return finish_predict(net, L, B, STEP_RESIDENT, p, v, z);
}

// Lane of a pipelined train call, chosen without locking: lane 0 in synchronous mode, round robin over the Hogwild lanes
TrainLane* pick_train_lane(ga3c_net* net) {
  if (!net->hogwild || net->xtr.empty()) return &net->tr;
  const size_t n = net->xtr.size() + 1;
  const size_t i = net->trr.fetch_add(1) % n;
  return i == 0 ? &net->tr : net->xtr[i - 1];
}

// A free intake of lane t, locked: whichever is not being staged / trained by another thread, else wait for one in turn.
Intake* take_intake(TrainLane& t) {
  const unsigned start = t.in_rr.fetch_add(1);
  for (unsigned k = 0; k < 2; ++k) {
    Intake* c = &t.in[(start + k) & 1];
    if (c->mu.try_lock()) return c;
  }
  Intake* c = &t.in[start & 1];
  c->mu.lock();
  return c;
}

// Resident / development entry points work on lane t's bound buffers directly: nobody may be staging into them meanwhile.
struct ResidentHold {
  TrainLane& t;
  explicit ResidentHold(TrainLane& lane) : t(lane) {
    t.in[0].mu.lock();
    t.in[1].mu.lock();
    t.mu.lock();
  }
  ~ResidentHold() {
    t.mu.unlock();
    t.in[1].mu.unlock();
    t.in[0].mu.unlock();
  }
  ResidentHold(const ResidentHold&) = delete;
  ResidentHold& operator=(const ResidentHold&) = delete;
};

// One call on a batch that arrives from outside: `stage` puts it into a free intake on the lane's staging stream (the
// calling thread owns that intake for the whole call); then, in the lane's turn, `body` runs with the intake bound to the
// lane and the train stream ordered behind the staging.  While one thread is inside `body` (a step in flight), the next
// thread's `stage` -- the PCIe gather of its rows -- proceeds on the staging stream.
// in_line: the staging is HBM to HBM and short (rows out of the state cache): it goes on the TRAIN stream itself, in front of
// the step -- beside a step, on the staging stream, even a 10-us copy stretched the step's kernels (conv2_dw 9 -> 43 us).
template <class StageFn, class BodyFn, class DoneFn>
int with_staged_batch(ga3c_net* net, int B, StageFn&& stage, BodyFn&& body, DoneFn&& done, bool in_line = false) {
  if (B < 1 || B > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", B, net->maxB);
  net->resident_gen.fetch_add(1, std::memory_order_relaxed);
  HIPCHK(hipSetDevice(net->cfg.device));
  TrainLane* t = pick_train_lane(net);
  const int64_t t0 = now_ns();
  roctxRangePushA("ga3c.train.stage");
  struct PopOnce { bool live = true; void pop() { if (live) { roctxRangePop(); live = false; } } ~PopOnce() { pop(); } } stage_range;
  Intake* in = take_intake(*t);
  std::lock_guard<std::mutex> ig(in->mu, std::adopt_lock);
  Stage s = intake_stage(*t, *in);
  if (in_line) s.st = t->st;
  CHK(stage(s));
  in->x_u8 = s.x_u8;
  if (!in_line) HIPCHK(hipEventRecord(in->ready, t->gst));
  stage_range.pop();
  const int64_t t1 = now_ns();
  {
    // The lane goes from trainer thread to trainer thread while the steps follow each other on the GPU: a thread asleep in
    // the mutex takes ~10 us to wake, so it spins for a step's length first (~23 us of waiting per call in the running engine).
    bool got = false;
    for (int spin = 0; spin < 4000 && !(got = t->mu.try_lock()); ++spin) __builtin_ia32_pause();
    if (!got) t->mu.lock();
    std::lock_guard<std::mutex> tl(t->mu, std::adopt_lock);
    const int64_t t2 = now_ns();
    bind_intake(*t, *in);
    if (!in_line) HIPCHK(hipStreamWaitEvent(t->st, in->ready, 0));
    stat_add(net, GA3C_STAT_TRAIN_STAGE_NS, t1 - t0);
    stat_add(net, GA3C_STAT_TRAIN_LANE_WAIT_NS, t2 - t1);
    CHK(body(*t, *in));
  }
  // the lane is free again: the next thread enqueues ITS step right behind this one while this thread waits for its own
  return done(*t, *in);
}

// the body of every train entry point, under the lane's mutex: forward, backward, update ENQUEUED, an event behind them
int train_enqueue(ga3c_net* net, TrainLane& t, Intake& in, int B, float lr, float beta) {
  TraceRange range("ga3c.train.step");
  const int64_t t0 = now_ns();
  t.wrote = -1;
  CHK(train_grads(net, t, B, beta, true, lr));
  CHK(train_apply(net, t, lr));
  in.wbuf = net->hogwild ? -1 : t.wrote;
  in.wbuf_seq = t.wrote_seq;
  HIPCHK(hipEventRecord(in.done, t.st));
  stat_add(net, GA3C_STAT_TRAIN_CALLS, 1);
  stat_add(net, GA3C_STAT_TRAIN_ROWS, B);
  stat_add(net, GA3C_STAT_TRAIN_LAUNCH_NS, now_ns() - t0);
  return GA3C_OK;
}

// ... and, with the lane released: wait for THIS step, hand its weights to the predictions, return its losses
int train_finish(ga3c_net* net, Intake& in, float* losses) {
  const int64_t t0 = now_ns();
  HIPCHK(hipEventSynchronize(in.done));
  adopt_buffer(net, in.wbuf, in.wbuf_seq);
  if (losses) memcpy(losses, in.losses, 3 * sizeof(float));
  stat_add(net, GA3C_STAT_TRAIN_SYNC_NS, now_ns() - t0);
  return GA3C_OK;
}

int alloc_train_lane(ga3c_net* net, TrainLane& t, float* shared_grad) {
  const int maxB = net->maxB, A = net->A;
  // The train stream is a high-priority stream: the runtime keeps a pool of hardware queues per priority, so it gets a
  // queue that no prediction lane, no staging stream and not the null stream (which holds one of the four normal queues
  // as soon as a hipMemcpy has run) can be multiplexed onto -- measured with tools/qmap.hip; the staging stream created
  // as the fifth normal stream had landed on the train stream's queue and its gather ran behind the step it was meant to
  // overlap (two trainer threads: 161 us per call, as slow as one).  Its kernels are also the ones that need every CU.
  {
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const bool plain = getenv("GA3C_TRAIN_PRIORITY") && atoi(getenv("GA3C_TRAIN_PRIORITY")) == 0;
    HIPCHK(stream_take(net->cfg.device, !plain, &t.st));
  }
  HIPCHK(stream_take(net->cfg.device, false, &t.gst));
  CHK(alloc_fwd(t.f, maxB, A));
  for (int k = 0; k < 2; ++k) {
    Intake& in = t.in[k];
    if (k == 0) {                      // intake 0 takes over the x buffers alloc_fwd made
      in.x = t.f.x;
      in.xu8 = t.f.xu8;
    } else {
      CHK(dmalloc(&in.x, (size_t)maxB * XS));
      HIPCHK(hipMalloc((void**)&in.xu8, (size_t)maxB * XS));
    }
    CHK(dmalloc(&in.yr, maxB));
    CHK(dmalloc(&in.act, (size_t)maxB * A));
    HIPCHK(hipHostMalloc((void**)&in.h_in, ((size_t)maxB * (XS + 1 + A)) * sizeof(float), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&in.h_off, (size_t)maxB * sizeof(int64_t), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&in.ready, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&in.done, hipEventDisableTiming));
    // the three loss sums are written by the kernel that completes them straight into pinned host memory (as p and v of a
    // prediction are): no copy kernel behind the step
    HIPCHK(hipHostMalloc((void**)&in.losses, 4 * sizeof(float), hipHostMallocDefault));
    memset(in.losses, 0, 4 * sizeof(float));
  }
  bind_intake(t, t.in[0]);
  CHK(dmalloc(&t.dz, (size_t)maxB * A));
  CHK(dmalloc(&t.dv, maxB));
  CHK(dmalloc(&t.lossrow, (size_t)maxB * 3));
  CHK(dmalloc(&t.dd1, (size_t)maxB * HID));
  CHK(dmalloc(&t.dn2, (size_t)maxB * FLAT));
  CHK(dmalloc(&t.dn1, (size_t)maxB * N1S));
  CHK(dmalloc(&t.slab2, (size_t)256 * SLAB2));   // conv2_dw: at most 256 sample groups
  CHK(dmalloc(&t.slab1, (size_t)512 * SLAB1));   // conv1_dw: at most 512 workgroups
  CHK(dmalloc(&t.scales, 16));
  if (shared_grad) {
    t.grad = shared_grad;
  } else {
    CHK(dmalloc(&t.grad, (size_t)net->n));
    t.owns_grad = true;
  }
  HIPCHK(hipHostMalloc((void**)&t.h_out, ((size_t)maxB * (2 * A + 1) + 4) * sizeof(float), hipHostMallocDefault));
  HIPCHK(hipEventCreate(&t.ev0));
  HIPCHK(hipEventCreate(&t.ev1));
  return GA3C_OK;
}

void free_train_lane(TrainLane& t) {
  if (!t.in[0].x && !t.in[0].xu8) {    // allocation failed before intake 0 took alloc_fwd's x buffers over
    if (t.f.x) (void)hipFree(t.f.x);
    if (t.f.xu8) (void)hipFree(t.f.xu8);
  }
  for (Intake& in : t.in) {            // the x buffers, y / a and the pinned arrays belong to the intakes
    if (in.x) (void)hipFree(in.x);
    if (in.xu8) (void)hipFree(in.xu8);
    if (in.yr) (void)hipFree(in.yr);
    if (in.act) (void)hipFree(in.act);
    if (in.h_in) (void)hipHostFree(in.h_in);
    if (in.h_off) (void)hipHostFree(in.h_off);
    if (in.ready) (void)hipEventDestroy(in.ready);
    if (in.done) (void)hipEventDestroy(in.done);
    if (in.losses) (void)hipHostFree(in.losses);
    in.x = nullptr; in.xu8 = nullptr;
  }
  t.f.x = nullptr;
  t.f.xu8 = nullptr;
  free_fwd(t.f);
  for (float* p : {t.dz, t.dv, t.lossrow, t.dd1, t.dn2, t.dn1, t.slab2, t.slab1, t.scales})
    if (p) (void)hipFree(p);
  if (t.owns_grad && t.grad) (void)hipFree(t.grad);
  if (t.h_out) (void)hipHostFree(t.h_out);
  if (t.gst) stream_give_back(t.gst);
  if (t.ev0) (void)hipEventDestroy(t.ev0);
  if (t.ev1) (void)hipEventDestroy(t.ev1);
  if (t.st) stream_give_back(t.st);
}

// (a stream with nothing pending answers hipStreamQuery at once; hipStreamSynchronize on it costs several microseconds
// all the same -- five idle streams were 30 us of a bracketed 20-step block)
static hipError_t sync_stream(hipStream_t st) {
  const hipError_t q = hipStreamQuery(st);
  if (q == hipSuccess) return hipSuccess;
  if (q != hipErrorNotReady) return q;
  (void)hipGetLastError();                           // "not ready" is an answer, not a failure: it must not surface in a later hipGetLastError
  return hipStreamSynchronize(st);
}

int sync_all(ga3c_net* net) {
  for (size_t i = 0; i < net->lanes.size(); ++i)             // (lanes beyond the lane streams borrow one: each stream once)
    if (net->lanes[i]->owns_st) HIPCHK(sync_stream(net->lanes[i]->st));
  HIPCHK(sync_stream(net->tr.st));
  for (TrainLane* t : net->xtr) HIPCHK(sync_stream(t->st));
  {
    std::lock_guard<std::mutex> g(net->ready_mu);   // nothing is in flight: the newest weights are complete
    net->cur.store(net->latest);   // (event_valid is irrelevant while cur == latest)
  }
  return GA3C_OK;
}

void lane_driver_main(ga3c_net* net, int l) {
  LaneDrivers& d = net->drv;
  const bool dev_ok = hipSetDevice(net->cfg.device) == hipSuccess;
  uint64_t seen = 0;
  for (;;) {
    int batch, iters, nlanes, idx;
    {
      // blocks of a benchmark follow each other within a fraction of a millisecond: stay awake that long, then sleep
      const int64_t until = now_ns() + 2000000;
      while (d.seq_hint.load(std::memory_order_acquire) == seen && now_ns() < until) __builtin_ia32_pause();
      std::unique_lock<std::mutex> lk(d.mu);
      d.cv.wait(lk, [&] { return d.quit || d.seq != seen; });
      if (d.quit) return;
      seen = d.seq;
      batch = d.batch; iters = d.iters; nlanes = d.nlanes; idx = d.idx;
    }
    if (l >= nlanes) continue;                                   // this block runs on fewer lanes
    d.ready.fetch_add(1, std::memory_order_acq_rel);
    while (d.go.load(std::memory_order_acquire) != seen) __builtin_ia32_pause();
    Lane* L = net->lanes[l];
    if (!dev_ok) {
      d.rcs[l] = GA3C_EHIP;
      d.errs[l] = "hipSetDevice failed on the lane's driver thread";
    } else {
      PredictInFlight inflight(net);
      (void)hipEventRecord(L->tm0, L->st);
      for (int i = l; i < iters; i += nlanes) {                  // lane l takes the steps l, l + nlanes, ...
        const int rc = lane_step(net, *L, idx, batch, STEP_RESIDENT, nullptr, nullptr);
        if (rc != GA3C_OK) { d.rcs[l] = rc; d.errs[l] = g_err; break; }
      }
      if (hipEventRecord(L->tm1, L->st) != hipSuccess || hipEventSynchronize(L->tm1) != hipSuccess) {
        if (d.rcs[l] == GA3C_OK) {
          d.rcs[l] = GA3C_EHIP;
          d.errs[l] = "waiting for the lane's last step failed";
        }
      }
    }
    d.done.fetch_add(1, std::memory_order_acq_rel);
  }
}

void stop_lane_drivers(ga3c_net* net) {
  LaneDrivers& d = net->drv;
  {
    std::lock_guard<std::mutex> lk(d.mu);
    d.quit = true;
  }
  d.cv.notify_all();
  for (auto& t : d.th)
    if (t.joinable()) t.join();
  d.th.clear();
}

}  // namespace

extern "C" {

const char* ga3c_last_error(void) { return g_err.c_str(); }

int ga3c_device_count(int32_t* count) {
  if (!count) return fail(GA3C_EINVAL, "null argument");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  *count = n;
  return GA3C_OK;
}

int ga3c_device_pci_bus_id(int32_t device, char* out, int32_t len) {
  if (!out || len < 16) return fail(GA3C_EINVAL, "bad argument");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(GA3C_EINVAL, "device %d not in [0,%d)", device, n);
  HIPCHK(hipDeviceGetPCIBusId(out, len, device));
  return GA3C_OK;
}

int ga3c_net_create(const ga3c_net_config* cfg, ga3c_net** out) {
  if (!cfg || !out) return fail(GA3C_EINVAL, "null argument");
  if (cfg->num_actions < 1 || cfg->num_actions > GA3C_MAX_ACTIONS)
    return fail(GA3C_EINVAL, "num_actions %d outside [1,%d]", cfg->num_actions, GA3C_MAX_ACTIONS);
  if (cfg->max_batch < 1 || cfg->max_batch > 65536) return fail(GA3C_EINVAL, "max_batch %d outside [1,65536]", cfg->max_batch);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(GA3C_EINVAL, "device %d not in [0,%d)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  // How a host thread waits for its stream: HIP's default spins (lowest latency, one core per waiting thread); with more
  // batching threads than the process has cores to spare, GA3C_BLOCKING_SYNC=1 makes them sleep on the completion interrupt.
  if (const char* e = getenv("GA3C_BLOCKING_SYNC")) {
    if (atoi(e) != 0 && hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
  }
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(GA3C_ESTATE, "device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName);
  ga3c_net* net = new (std::nothrow) ga3c_net();
  if (!net) return fail(GA3C_EINVAL, "out of host memory");
  net->cfg = *cfg;
  for (auto& c : net->stat) c.store(0);
  for (auto& c : net->stream_busy) c.store(0);
  net->fused_conv = getenv("GA3C_SPLIT_CONV") == nullptr;
  net->graphs = getenv("GA3C_GRAPHS") != nullptr;
  if (const char* e = getenv("GA3C_D1B_TILE_MAX")) net->d1b_tile_max = atoi(e);
  if (const char* e = getenv("GA3C_D1B_TAIL")) net->d1b_tail = atoi(e) != 0;
  if (const char* e = getenv("GA3C_D1F_TILE")) net->d1f_tile = atoi(e) != 0;
  if (const char* e = getenv("GA3C_D1F_FRAG_LANES")) net->d1f_frag_lanes = atoi(e);
  if (const char* e = getenv("GA3C_CONV_BWD")) net->conv_bwd_fused = atoi(e) != 0;
  if (const char* e = getenv("GA3C_CONV_BWD_MIN")) net->conv_bwd_min = atoi(e);
  if (const char* e = getenv("GA3C_C2F_QUARTER")) net->c2f_quarter = atoi(e) != 0;
  if (const char* e = getenv("GA3C_C2DW_OCC")) net->c2dw_occ = atoi(e);
  if (const char* e = getenv("GA3C_WD_STEP_IN_CONV_BWD")) net->wd_step_in_conv_bwd = atoi(e);
  if (const char* e = getenv("GA3C_WD_STEP_IN_CONV2_DX")) net->wd_step_in_conv2_dx = atoi(e) != 0;
  if (const char* e = getenv("GA3C_WD_BLOCKS_FIRST")) net->wd_blocks_first = atoi(e);
  if (const char* e = getenv("GA3C_DW_PAIR")) net->dw_pair = atoi(e) != 0;
  if (const char* e = getenv("GA3C_C1DW_BLOCKS")) net->c1dw_blocks = atoi(e) >= 1 && atoi(e) <= 512 ? atoi(e) : 0;
  if (const char* e = getenv("GA3C_TIME_PREDICTIONS")) net->time_predictions = atoi(e) != 0;
  if (const char* e = getenv("GA3C_OFFSETS_IN_ARGS")) net->offsets_in_args = atoi(e) != 0;
  if (const char* e = getenv("GA3C_STOP_EVENTS")) net->stop_events = atoi(e) != 0;
  if (const char* e = getenv("GA3C_FRAMES_IN_LINE")) net->frames_in_line = atoi(e) != 0;
  if (const char* e = getenv("GA3C_FUSED_UPDATE")) net->fused_update = atoi(e) != 0;
  if (const char* e = getenv("GA3C_GATHER_BLOCKS")) net->gather_max_blocks = atoi(e) > 0 ? atoi(e) : 32;
  for (const void* fn : {reinterpret_cast<const void*>(&conv_bwd_kernel<true, false>), reinterpret_cast<const void*>(&conv_bwd_kernel<false, false>),
                         reinterpret_cast<const void*>(&conv_bwd_kernel<true, true>), reinterpret_cast<const void*>(&conv_bwd_kernel<false, true>)}) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CB_LDS_FLOATS * sizeof(float)));
    if (e != hipSuccess) {
      delete net;
      return fail(GA3C_EHIP, "cannot reserve LDS for conv_bwd_kernel: %s", hipGetErrorString(e));
    }
  }
  const void* d1f_fns[2] = {reinterpret_cast<const void*>(&dense1_fwd_tile_kernel<1>), reinterpret_cast<const void*>(&dense1_fwd_tile_kernel<2>)};
  for (int i = 0; i < 2; ++i) {
    const void* fn = d1f_fns[i];
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      delete net;
      return fail(GA3C_EHIP, "cannot reserve LDS for dense1_fwd_tile_kernel: %s", hipGetErrorString(e));
    }
  }
  for (const void* fn : {reinterpret_cast<const void*>(&dense1_bwd_tile_kernel<0>), reinterpret_cast<const void*>(&dense1_bwd_tile_kernel<1>),
                         reinterpret_cast<const void*>(&dense1_bwd_tile_kernel<2>), reinterpret_cast<const void*>(&dense1_bwd_tile_kernel<0, true>),
                         reinterpret_cast<const void*>(&dense1_bwd_tile_kernel<1, true>), reinterpret_cast<const void*>(&dense1_bwd_tile_kernel<2, true>)}) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(D1B_LDS_FLOATS_TAIL * sizeof(float)));
    if (e != hipSuccess) {
      delete net;
      return fail(GA3C_EHIP, "cannot reserve LDS for dense1_bwd_tile_kernel: %s", hipGetErrorString(e));
    }
  }
  if (net->fused_conv) {
    const int lds = (int)(CS_LDS_FLOATS * sizeof(float));
    const void* fns[4] = {reinterpret_cast<const void*>(&conv_stack_fwd_kernel<true, true>),
                          reinterpret_cast<const void*>(&conv_stack_fwd_kernel<true, false>),
                          reinterpret_cast<const void*>(&conv_stack_fwd_kernel<false, true>),
                          reinterpret_cast<const void*>(&conv_stack_fwd_kernel<false, false>)};
    for (const void* fn : fns) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        delete net;
        return fail(GA3C_EHIP, "cannot reserve %d bytes of LDS for conv_stack_fwd_kernel: %s", lds, hipGetErrorString(e));
      }
    }
  }
  net->A = cfg->num_actions;
  net->maxB = cfg->max_batch;
  net->n = arena_floats(net->A);
  const int A = net->A, maxB = net->maxB;
  const int64_t offs[11] = {OFF_W1, OFF_B1, OFF_W2, OFF_B2, OFF_WD, OFF_BD, OFF_WV, OFF_BV, OFF_WP, off_bp(A), net->n};
  for (int i = 0; i < 11; ++i) net->tt.off[i] = offs[i];
#define TRY(expr)                    \
  do {                               \
    int _r = (expr);                 \
    if (_r != GA3C_OK) {             \
      std::string keep = g_err;      \
      ga3c_net_destroy(net);         \
      g_err = keep;                  \
      return _r;                     \
    }                                \
  } while (0)
#define TRYHIP(expr)                                                           \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      std::string keep = std::string(#expr) + " failed: " + hipGetErrorString(_e); \
      ga3c_net_destroy(net);                                                   \
      g_err = keep;                                                            \
      return GA3C_EHIP;                                                        \
    }                                                                          \
  } while (0)
  for (int i = 0; i < ga3c_net::NBUF; ++i) {
    TRY(dmalloc(&net->theta[i], (size_t)net->n));
    TRY(dmalloc(&net->theta_pk[i], (size_t)PK_FLOATS));
    TRYHIP(hipEventCreateWithFlags(&net->theta_ready[i], hipEventDisableTiming));
  }
  TRY(dmalloc(&net->grad, (size_t)net->n));
  TRY(dmalloc(&net->ms, (size_t)net->n));
  TRY(dmalloc(&net->mom, (size_t)net->n));
  {   // RMSProp `ms` slot starts at ones (TF RMSPropOptimizer._create_slots)
    std::vector<float> ones((size_t)net->n, 1.0f);
    TRYHIP(hipMemcpy(net->ms, ones.data(), (size_t)net->n * sizeof(float), hipMemcpyHostToDevice));
  }
  int nl = cfg->predict_lanes > 0 ? cfg->predict_lanes : 2;
  if (nl > 16) nl = 16;
  // Streams.  The runtime multiplexes every HIP stream of the process onto GPU_MAX_HW_QUEUES hardware queues (4 unless the
  // environment says otherwise), a new stream taking the least-used queue, and the chip serves four compute queues at a
  // time (tools/qmap.hip; profiles/README.md, "Hardware queues").  A prediction lane that lands on the train stream's
  // queue waits behind whole train steps, and a fifth busy queue makes all of them take turns -- either way the train step
  // is what slows down, and the engine's throughput is tied to it (every row served is trained once).  So the engine
  // keeps to FOUR normal-priority streams: three for the prediction lanes, created first, and the train lane's staging
  // stream (the train stream itself is a high-priority stream with a queue of its own, alloc_train_lane; with the state cache
  // or the frame queue on the device the staging is in line on it and the fourth stream stays idle).  Lanes beyond three
  // (Config.PREDICTORS > 3, the dynamic adjustment walking NP up) share those three streams in turn -- what the hardware
  // queue would do to them anyway.  Round 3 kept to two lane streams: NP = 3 then ran slower than NP = 2 (6.98 against 7.98 M
  // resident predictions/s; 9.33 M with a stream each), and ThreadDynamicAdjustment's random walk passes through it; in the
  // running engine three streams are worth +9-12 % with 3 or 4 predictor threads (64 Python agents: 460 -> 511 k and 428 ->
  // 480 k predictions/s) and nothing with 2 (profiles/r04_engine_matrix.md, call h).
  net->lane_streams = 3;
  if (const char* e = getenv("GA3C_LANE_STREAMS")) net->lane_streams = atoi(e) > 0 ? atoi(e) : 3;
  if (net->graphs) net->lane_streams = 16;     // a stream under capture cannot be shared
  for (int i = 0; i < nl; ++i) {
    Lane* L = new (std::nothrow) Lane();
    if (!L) { ga3c_net_destroy(net); return fail(GA3C_EINVAL, "out of host memory"); }
    net->lanes.push_back(L);
    L->sidx = i % net->lane_streams;
    if (i < net->lane_streams) {
      TRYHIP(stream_take(cfg->device, false, &L->st));
    } else {
      Lane* host = net->lanes[(size_t)(i % net->lane_streams)];
      L->st = host->st;
      L->owns_st = false;
      L->shared_st = host->shared_st = true;
    }
    TRYHIP(hipEventCreateWithFlags(&L->done, hipEventDisableTiming));
    TRYHIP(hipEventCreate(&L->tm0));
    TRYHIP(hipEventCreate(&L->tm1));
    TRY(alloc_fwd(L->f, maxB, A));
    TRYHIP(hipHostMalloc((void**)&L->h_in, (size_t)maxB * XS * sizeof(float), hipHostMallocDefault));
    TRYHIP(hipHostMalloc((void**)&L->h_out, ((size_t)maxB * (2 * A + 1)) * sizeof(float), hipHostMallocDefault));
    TRYHIP(hipHostMalloc((void**)&L->h_off, (size_t)maxB * sizeof(int64_t), hipHostMallocDefault));
    TRYHIP(hipHostMalloc((void**)&L->cache_dst, (size_t)maxB * sizeof(int64_t), hipHostMallocDefault));
  }
  TRY(alloc_train_lane(net, net->tr, net->grad));
  net->hogwild = cfg->train_lanes >= 2;
  for (int i = 1; i < cfg->train_lanes && i < 8; ++i) {
    TrainLane* t = new (std::nothrow) TrainLane();
    if (!t) { ga3c_net_destroy(net); return fail(GA3C_EINVAL, "out of host memory"); }
    net->xtr.push_back(t);
    TRY(alloc_train_lane(net, *t, nullptr));
  }
  TRYHIP(hipDeviceSynchronize());
#undef TRY
#undef TRYHIP
  *out = net;
  return GA3C_OK;
}

int ga3c_net_destroy(ga3c_net* net) {
  if (!net) return GA3C_OK;
  stop_lane_drivers(net);
  (void)hipSetDevice(net->cfg.device);
  (void)hipDeviceSynchronize();
  if (net->cache_ring) (void)hipFree(net->cache_ring);
  if (net->comm) (void)ncclCommDestroy(net->comm);
  if (net->cst) stream_give_back(net->cst);
  for (hipEvent_t e : {net->ev_tail_ready, net->ev_head_ready, net->ev_comm_done})
    if (e) (void)hipEventDestroy(e);
  for (Lane* L : net->lanes) {
    drop_graphs(*L);
    free_fwd(L->f);
    if (L->h_in) (void)hipHostFree(L->h_in);
    if (L->h_out) (void)hipHostFree(L->h_out);
    if (L->h_off) (void)hipHostFree(L->h_off);
    if (L->cache_dst) (void)hipHostFree(L->cache_dst);
    for (hipEvent_t e : {L->done, L->tm0, L->tm1})
      if (e) (void)hipEventDestroy(e);
    if (L->st && L->owns_st) stream_give_back(L->st);
    delete L;
  }
  free_train_lane(net->tr);
  for (TrainLane* t : net->xtr) {
    free_train_lane(*t);
    delete t;
  }
  free_frames(net->fr);
  if (net->reg_host) (void)hipHostUnregister(net->reg_host);
  for (int i = 0; i < ga3c_net::NBUF; ++i) {
    if (net->theta[i]) (void)hipFree(net->theta[i]);
    if (net->theta_pk[i]) (void)hipFree(net->theta_pk[i]);
    if (net->theta_ready[i]) (void)hipEventDestroy(net->theta_ready[i]);
  }
  for (float* p : {net->grad, net->ms, net->mom})
    if (p) (void)hipFree(p);
  delete net;
  return GA3C_OK;
}

int ga3c_net_param_count(ga3c_net* net, int64_t* count) {
  if (!net || !count) return fail(GA3C_EINVAL, "null argument");
  *count = net->n;
  return GA3C_OK;
}

static float* arena_ptr(ga3c_net* net, int which) {
  switch (which) {
    case 0: return net->theta[net->latest];
    case 1: return net->ms;
    case 2: return net->mom;
    case 3: return net->grad;
    default: return nullptr;
  }
}

int ga3c_net_get_arena(ga3c_net* net, int32_t which, float* out, int64_t count) {
  if (!net || !out) return fail(GA3C_EINVAL, "null argument");
  if (count != net->n) return fail(GA3C_EINVAL, "count %lld != arena size %lld", (long long)count, (long long)net->n);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  std::vector<std::unique_lock<std::mutex>> xl;   // Hogwild: no train lane may be mid-step while the arena is copied
  for (TrainLane* t : net->xtr) xl.emplace_back(t->mu);
  std::unique_lock<std::shared_mutex> lk(net->wmu);
  CHK(sync_all(net));
  float* src = arena_ptr(net, which);
  if (!src) return fail(GA3C_EINVAL, "arena selector %d not in [0,3]", which);
  HIPCHK(hipMemcpy(out, src, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
  return GA3C_OK;
}

int ga3c_net_set_arena(ga3c_net* net, int32_t which, const float* in, int64_t count) {
  if (!net || !in) return fail(GA3C_EINVAL, "null argument");
  if (count != net->n) return fail(GA3C_EINVAL, "count %lld != arena size %lld", (long long)count, (long long)net->n);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  std::vector<std::unique_lock<std::mutex>> xl;
  for (TrainLane* t : net->xtr) xl.emplace_back(t->mu);
  std::unique_lock<std::shared_mutex> lk(net->wmu);
  CHK(sync_all(net));
  float* dst = arena_ptr(net, which);
  if (!dst) return fail(GA3C_EINVAL, "arena selector %d not in [0,3]", which);
  HIPCHK(hipMemcpy(dst, in, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
  if (which == 0) {
    hipLaunchKernelGGL(pack_wd_kernel, dim3(KSTEPS_DENSE), dim3(256), 0, net->tr.st, net->theta[net->latest] + OFF_WD,
                       net->theta_pk[net->latest]);
    hipLaunchKernelGGL(pack_conv_kernel, dim3(48), dim3(256), 0, net->tr.st, net->theta[net->latest], net->theta_pk[net->latest]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(net->tr.st));
  }
  return GA3C_OK;
}

// ---- the variables by name (TensorFlow variable names, NetworkDNav.py:81-90, NetworkVP_discrate.py:60,63)
namespace {
constexpr int NPARAMS = 10;
const char* const PARAM_NAMES[NPARAMS] = {"conv11/w", "conv11/b", "conv12/w", "conv12/b", "dense1/w", "dense1/b",
                                          "logits_v/w", "logits_v/b", "logits_p/w", "logits_p/b"};
struct ParamInfo { int64_t off, count; int ndim; int64_t shape[4]; };
bool param_lookup(ga3c_net* net, const char* name, ParamInfo* pi) {
  const int A = net->A;
  const int64_t offs[NPARAMS + 1] = {OFF_W1, OFF_B1, OFF_W2, OFF_B2, OFF_WD, OFF_BD, OFF_WV, OFF_BV, OFF_WP, off_bp(A), net->n};
  const int64_t shapes[NPARAMS][4] = {{8, 8, 4, 16}, {16, 0, 0, 0}, {4, 4, 16, 32}, {32, 0, 0, 0}, {FLAT, HID, 0, 0}, {HID, 0, 0, 0},
                                      {HID, 1, 0, 0}, {1, 0, 0, 0}, {HID, A, 0, 0}, {A, 0, 0, 0}};
  const int ndims[NPARAMS] = {4, 1, 4, 1, 2, 1, 2, 1, 2, 1};
  std::string key(name ? name : "");
  if (key.size() > 2 && key.compare(key.size() - 2, 2, ":0") == 0) key.resize(key.size() - 2);
  for (int i = 0; i < NPARAMS; ++i)
    if (key == PARAM_NAMES[i]) {
      pi->off = offs[i]; pi->count = offs[i + 1] - offs[i]; pi->ndim = ndims[i];
      for (int d = 0; d < 4; ++d) pi->shape[d] = shapes[i][d];
      return true;
    }
  return false;
}
}  // namespace

int32_t ga3c_net_num_params(ga3c_net* net) { return net ? NPARAMS : 0; }

const char* ga3c_net_param_name(ga3c_net* net, int32_t index) {
  return (net && index >= 0 && index < NPARAMS) ? PARAM_NAMES[index] : nullptr;
}

int ga3c_net_param_info(ga3c_net* net, const char* name, int64_t* offset, int64_t* count, int32_t* ndim, int64_t shape[4]) {
  if (!net || !name) return fail(GA3C_EINVAL, "null argument");
  ParamInfo pi;
  if (!param_lookup(net, name, &pi)) return fail(GA3C_EINVAL, "no variable named %s", name);
  if (offset) *offset = pi.off;
  if (count) *count = pi.count;
  if (ndim) *ndim = pi.ndim;
  if (shape) for (int d = 0; d < 4; ++d) shape[d] = pi.shape[d];
  return GA3C_OK;
}

// one variable's slice of an arena, under the locks ga3c_net_get_arena / set_arena take (nothing in flight meanwhile)
static int param_copy(ga3c_net* net, const char* name, int which, float* out, const float* in, int64_t count) {
  if (!net || !name || (!out && !in)) return fail(GA3C_EINVAL, "null argument");
  ParamInfo pi;
  if (!param_lookup(net, name, &pi)) return fail(GA3C_EINVAL, "no variable named %s", name);
  if (count != pi.count) return fail(GA3C_EINVAL, "%s has %lld elements, not %lld", name, (long long)pi.count, (long long)count);
  if (which < 0 || which > (in ? 2 : 3)) return fail(GA3C_EINVAL, "arena selector %d not in [0,%d]", which, in ? 2 : 3);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  std::vector<std::unique_lock<std::mutex>> xl;
  for (TrainLane* t : net->xtr) xl.emplace_back(t->mu);
  std::unique_lock<std::shared_mutex> lk(net->wmu);
  CHK(sync_all(net));
  float* arena = arena_ptr(net, which);
  if (out) {
    HIPCHK(hipMemcpy(out, arena + pi.off, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return GA3C_OK;
  }
  HIPCHK(hipMemcpy(arena + pi.off, in, (size_t)count * sizeof(float), hipMemcpyHostToDevice));
  if (which == 0) {                      // the packed copies of dense1/w and the conv filters follow the weights
    hipLaunchKernelGGL(pack_wd_kernel, dim3(KSTEPS_DENSE), dim3(256), 0, net->tr.st, net->theta[net->latest] + OFF_WD,
                       net->theta_pk[net->latest]);
    hipLaunchKernelGGL(pack_conv_kernel, dim3(48), dim3(256), 0, net->tr.st, net->theta[net->latest], net->theta_pk[net->latest]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(net->tr.st));
  }
  return GA3C_OK;
}

int ga3c_net_get_param(ga3c_net* net, const char* name, int32_t which, float* out, int64_t count) {
  if (!out) return fail(GA3C_EINVAL, "null argument");
  return param_copy(net, name, which, out, nullptr, count);
}

int ga3c_net_set_param(ga3c_net* net, const char* name, int32_t which, const float* in, int64_t count) {
  if (!in) return fail(GA3C_EINVAL, "null argument");
  return param_copy(net, name, which, nullptr, in, count);
}

int ga3c_net_save(ga3c_net* net, const char* path) {
  if (!net || !path) return fail(GA3C_EINVAL, "null argument");
  std::vector<float> arena[3];
  for (int w = 0; w < 3; ++w) {
    arena[w].resize((size_t)net->n);
    CHK(ga3c_net_get_arena(net, w, arena[w].data(), net->n));
  }
  const char* suffix[3] = {":0", "/RMSProp:0", "/RMSProp_1:0"};
  std::vector<ga3c_ckpt::Member> members;
  ga3c_ckpt::Member st;
  st.name = "step"; st.descr = "<i8";
  const int64_t step = net->step.load();
  st.bytes.assign(reinterpret_cast<const uint8_t*>(&step), reinterpret_cast<const uint8_t*>(&step) + 8);
  members.push_back(st);
  for (int i = 0; i < NPARAMS; ++i) {
    ParamInfo pi;
    param_lookup(net, PARAM_NAMES[i], &pi);
    for (int w = 0; w < 3; ++w) {
      ga3c_ckpt::Member m;
      m.name = std::string(PARAM_NAMES[i]) + suffix[w];
      m.descr = "<f4";
      m.shape.assign(pi.shape, pi.shape + pi.ndim);
      const uint8_t* src = reinterpret_cast<const uint8_t*>(arena[w].data() + pi.off);
      m.bytes.assign(src, src + (size_t)pi.count * sizeof(float));
      members.push_back(std::move(m));
    }
  }
  std::string err;
  if (!ga3c_ckpt::write_npz(path, members, &err)) return fail(GA3C_ESTATE, "%s", err.c_str());
  return GA3C_OK;
}

int ga3c_net_load(ga3c_net* net, const char* path) {
  if (!net || !path) return fail(GA3C_EINVAL, "null argument");
  std::map<std::string, ga3c_ckpt::Member> members;
  std::string err;
  if (!ga3c_ckpt::read_npz(path, &members, &err)) return fail(GA3C_ESTATE, "%s", err.c_str());
  const char* suffix[3] = {":0", "/RMSProp:0", "/RMSProp_1:0"};
  std::vector<float> arena[3];
  for (int w = 0; w < 3; ++w) arena[w].resize((size_t)net->n);
  for (int i = 0; i < NPARAMS; ++i) {
    ParamInfo pi;
    param_lookup(net, PARAM_NAMES[i], &pi);
    for (int w = 0; w < 3; ++w) {
      const std::string key = std::string(PARAM_NAMES[i]) + suffix[w];
      auto it = members.find(key);
      if (it == members.end()) return fail(GA3C_ESTATE, "%s holds no %s", path, key.c_str());
      const ga3c_ckpt::Member& m = it->second;
      int64_t elems = 1;
      for (int64_t d : m.shape) elems *= d;
      if (m.descr != "<f4" || elems != pi.count || m.bytes.size() != (size_t)pi.count * sizeof(float))
        return fail(GA3C_ESTATE, "%s: %s is %s with %lld elements, this network wants <f4 with %lld", path, key.c_str(),
                    m.descr.c_str(), (long long)elems, (long long)pi.count);
      memcpy(arena[w].data() + pi.off, m.bytes.data(), m.bytes.size());
    }
  }
  auto st = members.find("step");
  if (st == members.end() || st->second.descr != "<i8" || st->second.bytes.size() != 8)
    return fail(GA3C_ESTATE, "%s holds no int64 step", path);
  int64_t step = 0;
  memcpy(&step, st->second.bytes.data(), 8);
  for (int w = 0; w < 3; ++w) CHK(ga3c_net_set_arena(net, w, arena[w].data(), net->n));
  net->step.store(step);
  return GA3C_OK;
}

int ga3c_net_get_step(ga3c_net* net, int64_t* step) {
  if (!net || !step) return fail(GA3C_EINVAL, "null argument");
  *step = net->step.load();
  return GA3C_OK;
}

int ga3c_net_set_step(ga3c_net* net, int64_t step) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  net->step.store(step);
  return GA3C_OK;
}

int ga3c_net_predict(ga3c_net* net, const float* x, int32_t batch, float* p, float* v, float* z) {
  return predict_common(net, x, false, batch, p, v, z);
}

int ga3c_net_predict_u8(ga3c_net* net, const uint8_t* x, int32_t batch, float* p, float* v, float* z) {
  return predict_common(net, x, true, batch, p, v, z);
}

int ga3c_net_compute_grads(ga3c_net* net, const float* x, const float* y_r, const float* a, int32_t batch,
                           float beta, float* losses) {
  if (!net || !x || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  ResidentHold hold(net->tr);
  CHK(stage_train_inputs(net, net->tr, x, false, y_r, a, batch));
  CHK(train_grads(net, net->tr, batch, beta));
  return read_losses(net, net->tr, losses);
}

int ga3c_net_apply_grads(ga3c_net* net, float learning_rate) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  CHK(train_apply(net, net->tr, learning_rate));
  HIPCHK(hipStreamSynchronize(net->tr.st));
  return GA3C_OK;
}

int ga3c_net_train(ga3c_net* net, const float* x, const float* y_r, const float* a, int32_t batch,
                   float learning_rate, float beta, float* losses) {
  if (!net || !x || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  return with_staged_batch(net, batch, [&](Stage& s) { return stage_train_inputs(net, s, x, false, y_r, a, batch); },
                           [&](TrainLane& t, Intake& in) { return train_enqueue(net, t, in, batch, learning_rate, beta); },
                           [&](TrainLane&, Intake& in) { return train_finish(net, in, losses); });
}

int ga3c_net_train_u8(ga3c_net* net, const uint8_t* x, const float* y_r, const float* a, int32_t batch,
                      float learning_rate, float beta, float* losses) {
  if (!net || !x || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  return with_staged_batch(net, batch, [&](Stage& s) { return stage_train_inputs(net, s, x, true, y_r, a, batch); },
                           [&](TrainLane& t, Intake& in) { return train_enqueue(net, t, in, batch, learning_rate, beta); },
                           [&](TrainLane&, Intake& in) { return train_finish(net, in, losses); });
}

int ga3c_net_evaluate(ga3c_net* net, const float* x, const uint8_t* x_u8, const int64_t* offsets, int32_t offsets_u8,
                      const float* y_r, const float* a, int32_t batch, float beta, float* losses, float* d1, float* v,
                      float* p) {
  if (!net || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  if ((x != nullptr) + (x_u8 != nullptr) + (offsets != nullptr) != 1)
    return fail(GA3C_EINVAL, "evaluate: exactly one of x, x_u8, offsets names the states");
  return with_staged_batch(
      net, batch,
      [&](Stage& s) {
        if (offsets) return launch_gather(net, offsets, batch, offsets_u8 != 0, s, y_r, a);
        return stage_train_inputs(net, s, x ? (const void*)x : (const void*)x_u8, x == nullptr, y_r, a, batch);
      },
      [&](TrainLane& t, Intake&) { return evaluate_staged(net, t, batch, beta, losses, d1, v, p); },
      [](TrainLane&, Intake&) { return (int)GA3C_OK; });
}

int ga3c_net_register_host(ga3c_net* net, void* base, int64_t bytes) {
  if (!net || !base || bytes < 16) return fail(GA3C_EINVAL, "bad argument");
  if (net->reg_host) return fail(GA3C_ESTATE, "a host segment is already registered");
  HIPCHK(hipSetDevice(net->cfg.device));
  HIPCHK(hipHostRegister(base, (size_t)bytes, hipHostRegisterMapped));
  void* dev = nullptr;
  hipError_t e = hipHostGetDevicePointer(&dev, base, 0);
  if (e != hipSuccess) {
    (void)hipHostUnregister(base);
    return fail(GA3C_EHIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
  }
  drop_all_graphs(net);
  net->reg_host = base;
  net->reg_dev = (uint8_t*)dev;
  net->reg_bytes = bytes;
  return GA3C_OK;
}

int ga3c_net_unregister_host(ga3c_net* net) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  if (!net->reg_host) return GA3C_OK;
  HIPCHK(hipSetDevice(net->cfg.device));
  CHK(sync_all(net));
  drop_all_graphs(net);
  HIPCHK(hipHostUnregister(net->reg_host));
  net->reg_host = nullptr;
  net->reg_dev = nullptr;
  net->reg_bytes = 0;
  return GA3C_OK;
}

int ga3c_net_predict_gather(ga3c_net* net, const int64_t* offsets, int32_t batch, int32_t u8, float* p, float* v,
                            float* z) {
  if (!net || !offsets || !p || !v) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  PredictInFlight inflight(net);
  Lane* L = take_lane(net);
  LaneGuard guard(net, L);
  CHK(stage_offsets(net, offsets, batch, u8 != 0, L->h_off));
  return finish_predict(net, L, batch, u8 ? STEP_GATHER_U8 : STEP_GATHER_F32, p, v, z);
}

// rows' slots in the state cache: row i = (agents[i], seqs[i]) -> byte offset.  `store`: the rows are about to be STORED by a
// step (nothing is recorded yet: cache_commit does that once the step is enqueued); otherwise they are being READ and every
// slot must hold exactly the request it is named by, with room to spare before its agent's next predictions reach it
constexpr int CACHE_SLACK = 4;
static int cache_offsets(ga3c_net* net, const int32_t* agents, const int64_t* seqs, int batch, bool store, int64_t* out) {
  if (!net->cache_ring) return fail(GA3C_ESTATE, "no state cache configured (ga3c_net_state_cache_config)");
  std::lock_guard<std::mutex> g(net->cache_mu);
  for (int i = 0; i < batch; ++i) {
    const int ag = agents[i];
    if (ag < 0 || ag >= net->cache_agents) return fail(GA3C_EINVAL, "state cache: agent %d outside [0,%d)", ag, net->cache_agents);
    if (seqs[i] < 0) return fail(GA3C_EINVAL, "state cache: row %d has a negative request number", i);
    const int64_t slot = seqs[i] % net->cache_depth;
    if (!store) {
      const int64_t newest = net->cache_newest[(size_t)ag], tag = net->cache_tags[(size_t)ag * net->cache_depth + slot];
      if (tag != seqs[i] || newest - seqs[i] >= net->cache_depth - CACHE_SLACK) {
        stat_add(net, GA3C_STAT_STATE_CACHE_LOST, batch);
        return fail(GA3C_ELOST, "state cache: row %d, request %lld of agent %d, is not held (slot holds %lld, newest %lld, depth %d): "
                    "raise Config.STATE_CACHE_DEPTH", i, (long long)seqs[i], ag, (long long)tag, (long long)newest, net->cache_depth);
      }
    }
    out[i] = ((int64_t)ag * net->cache_depth + slot) * (int64_t)XS;
  }
  return GA3C_OK;
}

// the step that stores these rows has been enqueued: from now on the slots hold them (stream order puts any later copy
// of a slot behind the store)
static void cache_commit(ga3c_net* net, const int32_t* agents, const int64_t* seqs, int batch) {
  std::lock_guard<std::mutex> g(net->cache_mu);
  for (int i = 0; i < batch; ++i) {
    const int ag = agents[i];
    net->cache_tags[(size_t)ag * net->cache_depth + seqs[i] % net->cache_depth] = seqs[i];
    if (seqs[i] > net->cache_newest[(size_t)ag]) net->cache_newest[(size_t)ag] = seqs[i];
  }
}

static int predict_begin_common(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const int64_t* seqs, int32_t batch,
                                int32_t u8, int32_t* ticket) {
  if (!net || !offsets || !ticket) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  if (agents) {
    if (!seqs) return fail(GA3C_EINVAL, "null argument");
    if (!u8 || net->graphs)
      return fail(GA3C_ESTATE, "the state cache is filled by plain launches on uint8 states");
  }
  Lane* L = take_lane(net);                                  // stays taken until ga3c_net_predict_gather_end
  net->predict_inflight.fetch_add(1, std::memory_order_relaxed);
  auto give_back = [&]() {
    net->predict_inflight.fetch_sub(1, std::memory_order_relaxed);
    net->stream_busy[L->sidx].fetch_sub(1, std::memory_order_relaxed);
    L->mu.unlock();
  };
  int rc = stage_offsets(net, offsets, batch, u8 != 0, L->h_off);
  L->cache_on = false;
  if (rc == GA3C_OK && agents) {
    rc = cache_offsets(net, agents, seqs, batch, true, L->cache_dst);
    L->cache_on = rc == GA3C_OK;
  }
  if (rc == GA3C_OK) {
    TraceRange range("ga3c.predict.begin");
    const int64_t t0 = now_ns();
    float* hp = L->h_out;
    float* hv = hp + (size_t)net->maxB * net->A;
    // (lane_forward leaves L->done behind the step: _end only waits for it -- recorded by _end it was a round trip
    // through the queue of its own, after a step that had long finished)
    rc = lane_forward(net, *L, batch, u8 ? STEP_GATHER_U8 : STEP_GATHER_F32, hp, hv);
    stat_add(net, GA3C_STAT_PREDICT_LAUNCH_NS, now_ns() - t0);
    if (rc == GA3C_OK && L->cache_on) cache_commit(net, agents, seqs, batch);
  }
  L->cache_on = false;
  if (rc != GA3C_OK) {
    give_back();
    return rc;
  }
  for (size_t i = 0; i < net->lanes.size(); ++i)
    if (net->lanes[i] == L) *ticket = (int32_t)i;
  L->begun.store(true);
  return GA3C_OK;
}

int ga3c_net_predict_gather_begin(ga3c_net* net, const int64_t* offsets, int32_t batch, int32_t u8, int32_t* ticket) {
  return predict_begin_common(net, offsets, nullptr, nullptr, batch, u8, ticket);
}

int ga3c_net_predict_gather_begin_cached(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const int64_t* seqs,
                                         int32_t batch, int32_t u8, int32_t* ticket) {
  if (!agents || !seqs) return fail(GA3C_EINVAL, "null argument");
  return predict_begin_common(net, offsets, agents, seqs, batch, u8, ticket);
}

int ga3c_net_predict_gather_end(ga3c_net* net, int32_t ticket, int32_t batch, float* p, float* v) {
  if (!net || !p || !v) return fail(GA3C_EINVAL, "null argument");
  if (ticket < 0 || ticket >= (int32_t)net->lanes.size()) return fail(GA3C_EINVAL, "bad ticket %d", ticket);
  if (batch < 1 || batch > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", batch, net->maxB);
  Lane* L = net->lanes[(size_t)ticket];
  if (!L->begun.exchange(false)) return fail(GA3C_ESTATE, "ticket %d: no batch was begun on that lane (or it was ended already)", ticket);
  const int64_t t0 = now_ns();
  const hipError_t he = hipEventSynchronize(L->done);      // left behind the step by _begin
  const int rc = he == hipSuccess ? GA3C_OK : fail(GA3C_EHIP, "hipEventSynchronize failed: %s", hipGetErrorString(he));
  if (rc == GA3C_OK) {
    note_predict_span(net, L);
    const float* hp = L->h_out;
    const float* hv = hp + (size_t)net->maxB * net->A;
    memcpy(p, hp, (size_t)batch * net->A * sizeof(float));
    memcpy(v, hv, (size_t)batch * sizeof(float));
    stat_add(net, GA3C_STAT_PREDICT_CALLS, 1);
    stat_add(net, GA3C_STAT_PREDICT_ROWS, batch);
    stat_add(net, GA3C_STAT_PREDICT_SYNC_NS, now_ns() - t0);
  }
  net->predict_inflight.fetch_sub(1, std::memory_order_relaxed);
  net->stream_busy[L->sidx].fetch_sub(1, std::memory_order_relaxed);
  L->mu.unlock();
  return rc;
}

int ga3c_net_train_gather(ga3c_net* net, const int64_t* offsets, int32_t u8, const float* y_r, const float* a,
                          int32_t batch, float learning_rate, float beta, float* losses) {
  if (!net || !offsets || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  return with_staged_batch(
      net, batch,
      [&](Stage& s) { return launch_gather(net, offsets, batch, u8 != 0, s, y_r, a); },
      [&](TrainLane& t, Intake& in) { return train_enqueue(net, t, in, batch, learning_rate, beta); },
      [&](TrainLane&, Intake& in) { return train_finish(net, in, losses); });
}

// ---- frame front-end ------------------------------------------------------------------------------------------
int ga3c_net_frames_config(ga3c_net* net, int32_t max_agents, int32_t height, int32_t width, int32_t channels,
                           int32_t history) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  if (max_agents < 1 || height < 1 || width < 1 || (channels != 3 && channels != 4 && channels != 1))
    return fail(GA3C_EINVAL, "frames: need max_agents >= 1, a positive frame size and 3 or 4 channels (1: ready-made planes)");
  if (channels == 1 && (height != IMG || width != IMG))
    return fail(GA3C_EINVAL, "frames: one channel means ready-made %dx%d planes", IMG, IMG);
  if (history != 0 && history < 2 * CIN) return fail(GA3C_EINVAL, "frames: a plane history holds at least %d planes", 2 * CIN);
  HIPCHK(hipSetDevice(net->cfg.device));
  Frames& f = net->fr;
  drop_all_graphs(net);
  std::lock_guard<std::mutex> g(f.mu);
  free_frames(f);
  const ResampleTable th = make_bilinear_table(width, IMG), tv = make_bilinear_table(height, IMG);
  f.maxA = max_agents; f.H = height; f.W = width; f.C = channels; f.hks = th.ksize; f.vks = tv.ksize;
  f.frame_bytes = (size_t)height * width * channels;
  const int64_t npx = (int64_t)height * width;
  if (npx % 4 || npx > (int64_t)4 * FE_MAXG * FE_THREADS)
    return fail(GA3C_EINVAL, "frames: height*width must be a multiple of 4 and at most %d", 4 * FE_MAXG * FE_THREADS);
  if (th.ksize > FE_MAXK) return fail(GA3C_EINVAL, "frames: width %d needs %d taps per output, the kernel holds %d", width, th.ksize, FE_MAXK);
  f.lds = frontend_lds_bytes(height, width, IMG, IMG, f.hks, f.vks);
  if (f.lds > 150 * 1024) return fail(GA3C_EINVAL, "frames: a %dx%d frame does not fit the kernel's LDS plan", height, width);
  for (const void* fn : {reinterpret_cast<const void*>(&frame_frontend_kernel<3>),
                         reinterpret_cast<const void*>(&frame_frontend_kernel<4>)})
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)));
  std::vector<int32_t> tab;
  tab.insert(tab.end(), th.bounds.begin(), th.bounds.end());
  tab.insert(tab.end(), th.kk.begin(), th.kk.end());
  tab.insert(tab.end(), tv.bounds.begin(), tv.bounds.end());
  tab.insert(tab.end(), tv.kk.begin(), tv.kk.end());
  HIPCHK(hipMalloc((void**)&f.d_tab, tab.size() * sizeof(int32_t)));
  HIPCHK(hipMemcpy(f.d_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  f.hb = f.d_tab; f.hk = f.hb + th.bounds.size(); f.vb = f.hk + th.kk.size(); f.vk = f.vb + tv.bounds.size();
  const size_t words = (size_t)max_agents * IMG * IMG;
  HIPCHK(hipMalloc((void**)&f.stacks, words * sizeof(uint32_t)));
  HIPCHK(hipMemset(f.stacks, 0, words * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void**)&f.d_rgb, (size_t)max_agents * f.frame_bytes));
  HIPCHK(hipMalloc((void**)&f.d_planes, (size_t)max_agents * IMG * IMG));
  HIPCHK(hipHostMalloc((void**)&f.h_rgb, (size_t)max_agents * f.frame_bytes, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&f.h_planes, (size_t)max_agents * IMG * IMG, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&f.h_agents, (size_t)max_agents * sizeof(int32_t), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&f.h_reset, (size_t)max_agents, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&f.h_src, (size_t)max_agents * sizeof(int64_t), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&f.h_slot, (size_t)max_agents * sizeof(int32_t), hipHostMallocDefault));
  f.hist = history;
  if (history) HIPCHK(hipMalloc((void**)&f.ring, (size_t)max_agents * history * IMG * IMG));
  HIPCHK(stream_take(net->cfg.device, false, &f.st));
  f.filled.assign((size_t)max_agents, 0);
  f.pushed.assign((size_t)max_agents, 0);
  f.on = true;
  return GA3C_OK;
}

// rgb as the kernel can read it: in place when device-visible, else through the pinned staging buffer
static const uint8_t* frames_source(ga3c_net* net, const uint8_t* rgb, int n) {
  Frames& f = net->fr;
  const size_t bytes = (size_t)n * f.frame_bytes;
  const uint8_t* dev = device_visible(net, rgb, bytes);
  if (dev) return dev;
  memcpy(f.h_rgb, rgb, bytes);
  return f.h_rgb;
}

int ga3c_net_frames_preprocess(ga3c_net* net, const uint8_t* rgb, int32_t n, uint8_t* planes) {
  if (!net || !rgb || !planes) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (n < 1 || n > f.maxA) return fail(GA3C_EINVAL, "frames: %d frames outside [1,%d]", n, f.maxA);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> g(f.mu);
  CHK(launch_frames(net, frames_source(net, rgb, n), nullptr, nullptr, f.h_planes, n));   // planes land in pinned memory
  HIPCHK(hipStreamSynchronize(f.st));
  memcpy(planes, f.h_planes, (size_t)n * IMG * IMG);
  return GA3C_OK;
}

// common part of the two push entry points: validates the agent list, fills the pinned argument arrays, launches,
// waits, advances the host mirrors; seq_out[i] = sequence number of the plane agent i just got (counts from 0)
static int frames_push_core(ga3c_net* net, const uint8_t* rgb_dev, const int64_t* src_off, const int32_t* agents,
                            const uint8_t* reset, int n, int64_t* seq_out) {
  Frames& f = net->fr;
  std::vector<uint8_t> seen((size_t)f.maxA, 0);
  for (int i = 0; i < n; ++i) {
    if (agents[i] < 0 || agents[i] >= f.maxA) return fail(GA3C_EINVAL, "frames: agent %d outside [0,%d)", agents[i], f.maxA);
    if (seen[agents[i]]) return fail(GA3C_EINVAL, "frames: agent %d appears twice in one push", agents[i]);
    seen[agents[i]] = 1;
    f.h_agents[i] = agents[i];
    f.h_reset[i] = reset ? reset[i] : 0;
    f.h_slot[i] = f.hist ? (int32_t)(f.pushed[agents[i]] % f.hist) : 0;
    if (src_off) f.h_src[i] = src_off[i];
  }
  CHK(launch_frames(net, rgb_dev, f.h_agents, f.h_reset, nullptr, n, src_off ? f.h_src : nullptr));
  HIPCHK(hipStreamSynchronize(f.st));
  for (int i = 0; i < n; ++i) {
    int& d = f.filled[agents[i]];
    d = (reset && reset[i]) ? 1 : (d < CIN ? d + 1 : CIN);
    if (seq_out) seq_out[i] = f.pushed[agents[i]];
    f.pushed[agents[i]] += 1;
  }
  return GA3C_OK;
}

int ga3c_net_frames_push(ga3c_net* net, const uint8_t* rgb, const int32_t* agents, const uint8_t* reset, int32_t n,
                         int64_t* seq_out) {
  if (!net || !rgb || !agents) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (n < 1 || n > f.maxA) return fail(GA3C_EINVAL, "frames: %d frames outside [1,%d]", n, f.maxA);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> g(f.mu);
  return frames_push_core(net, frames_source(net, rgb, n), nullptr, agents, reset, n, seq_out);
}

int ga3c_net_frames_push_offsets(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const uint8_t* reset,
                                 int32_t n, int64_t* seq_out) {
  if (!net || !offsets || !agents) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (!net->reg_dev) return fail(GA3C_ESTATE, "no host segment registered (ga3c_net_register_host)");
  if (n < 1 || n > f.maxA) return fail(GA3C_EINVAL, "frames: %d frames outside [1,%d]", n, f.maxA);
  for (int i = 0; i < n; ++i)
    if (offsets[i] < 0 || offsets[i] + (int64_t)f.frame_bytes > net->reg_bytes || (offsets[i] & 3))
      return fail(GA3C_EINVAL, "frame %d: offset %lld outside the registered segment or not 4-byte aligned", i, (long long)offsets[i]);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> g(f.mu);
  return frames_push_core(net, net->reg_dev, offsets, agents, reset, n, seq_out);
}

int ga3c_net_frames_state(ga3c_net* net, int32_t agent, uint8_t* state, int32_t* filled) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (agent < 0 || agent >= f.maxA) return fail(GA3C_EINVAL, "frames: agent %d outside [0,%d)", agent, f.maxA);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> g(f.mu);
  if (filled) *filled = f.filled[agent];
  if (state) HIPCHK(hipMemcpy(state, f.stacks + (size_t)agent * IMG * IMG, XS, hipMemcpyDeviceToHost));
  return GA3C_OK;
}

int ga3c_net_predict_frames(ga3c_net* net, const int32_t* agents, int32_t n, float* p, float* v, float* z) {
  if (!net || !agents || !p || !v) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (n < 1 || n > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", n, net->maxB);
  HIPCHK(hipSetDevice(net->cfg.device));
  {
    std::lock_guard<std::mutex> g(f.mu);
    for (int i = 0; i < n; ++i) {
      if (agents[i] < 0 || agents[i] >= f.maxA) return fail(GA3C_EINVAL, "frames: agent %d outside [0,%d)", agents[i], f.maxA);
      if (f.filled[agents[i]] < CIN)
        return fail(GA3C_ESTATE, "frames: agent %d has %d of %d frames queued (Environment.py:64-65: no state yet)",
                    agents[i], f.filled[agents[i]], CIN);
    }
  }
  PredictInFlight inflight(net);
  Lane* L = take_lane(net);
  LaneGuard guard(net, L);
  for (int i = 0; i < n; ++i) L->h_off[i] = (int64_t)agents[i] * XS;
  return finish_predict(net, L, n, STEP_QUEUES, p, v, z);
}

// One batch of the frames predictor loop, in two halves: _begin pushes the batch's frames into the agents' queues and enqueues
// the forward pass for those that asked for one (the lane stays taken), _end waits for it and hands the answers out.  The
// native loop (ga3c_pq_serve_frames_pipelined) answers batch k between the two halves of batch k+1.
int ga3c_net_serve_frames_begin(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const uint32_t* flags, int32_t n,
                                int32_t* ticket) {
  if (!net || !offsets || !agents || !flags || !ticket) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (!net->reg_dev) return fail(GA3C_ESTATE, "no host segment registered (ga3c_net_register_host)");
  if (n < 1 || n > f.maxA || n > net->maxB) return fail(GA3C_EINVAL, "frames: %d requests outside [1,%d]", n, f.maxA < net->maxB ? f.maxA : net->maxB);
  HIPCHK(hipSetDevice(net->cfg.device));
  Lane* L = take_lane(net);                                  // stays taken until ga3c_net_serve_frames_end
  net->predict_inflight.fetch_add(1, std::memory_order_relaxed);
  auto give_back = [&]() {
    net->predict_inflight.fetch_sub(1, std::memory_order_relaxed);
    net->stream_busy[L->sidx].fetch_sub(1, std::memory_order_relaxed);
    L->mu.unlock();
  };
  // per-call argument arrays, carved out of the lane's (otherwise idle) pinned input staging and read by the kernels in place
  int32_t* h_ag = reinterpret_cast<int32_t*>(L->h_in);
  int32_t* h_slot = h_ag + net->maxB;
  int64_t* h_src = reinterpret_cast<int64_t*>(h_slot + net->maxB);
  uint8_t* h_reset = reinterpret_cast<uint8_t*>(h_src + net->maxB);
  int want = 0, rc = GA3C_OK;
  {
    std::lock_guard<std::mutex> g(f.mu);   // host mirrors of the queues; per agent the protocol allows one request in flight
    std::vector<uint8_t> seen((size_t)f.maxA, 0);
    for (int i = 0; i < n && rc == GA3C_OK; ++i) {
      const int a = agents[i];
      if (a < 0 || a >= f.maxA) rc = fail(GA3C_EINVAL, "frames: agent %d outside [0,%d)", a, f.maxA);
      else if (seen[a]) rc = fail(GA3C_EINVAL, "frames: agent %d appears twice in one batch", a);
      else if (offsets[i] < 0 || offsets[i] + (int64_t)f.frame_bytes > net->reg_bytes || (offsets[i] & 3))
        rc = fail(GA3C_EINVAL, "frame %d: offset %lld outside the registered segment or not 4-byte aligned", i, (long long)offsets[i]);
      else seen[a] = 1;
    }
    for (int i = 0; i < n && rc == GA3C_OK; ++i) {
      const int a = agents[i];
      const bool rs = (flags[i] & 1u) != 0;
      h_ag[i] = a; h_src[i] = offsets[i]; h_reset[i] = rs ? 1 : 0;
      h_slot[i] = f.hist ? (int32_t)(f.pushed[a] % f.hist) : 0;
      f.pushed[a] += 1;
      int& d = f.filled[a];
      d = rs ? 1 : (d < CIN ? d + 1 : CIN);
      if (!(flags[i] & 2u)) {
        if (d < CIN) rc = fail(GA3C_ESTATE, "frames: agent %d asks for a prediction with %d of %d frames queued", a, d, CIN);
        else L->h_off[want++] = (int64_t)a * XS;
      }
    }
  }
  if (rc == GA3C_OK) rc = launch_frames(net, net->reg_dev, h_ag, h_reset, nullptr, n, h_src, L->st, h_slot);   // same stream as the forward pass
  if (rc == GA3C_OK) {
    if (want == 0) {
      if (hipEventRecord(L->done, L->st) != hipSuccess) rc = fail(GA3C_EHIP, "hipEventRecord failed behind a batch of frames");
    } else {
      TraceRange range("ga3c.predict_frames");
      const int64_t t0 = now_ns();
      float* hp = L->h_out;
      rc = lane_forward(net, *L, want, STEP_QUEUES, hp, hp + (size_t)net->maxB * net->A);   // leaves L->done behind the step
      stat_add(net, GA3C_STAT_PREDICT_LAUNCH_NS, now_ns() - t0);
    }
  }
  if (rc != GA3C_OK) {
    give_back();
    return rc;
  }
  L->frames_want = want;
  for (size_t i = 0; i < net->lanes.size(); ++i)
    if (net->lanes[i] == L) *ticket = (int32_t)i;
  L->begun.store(true);
  return GA3C_OK;
}

int ga3c_net_serve_frames_end(ga3c_net* net, int32_t ticket, const uint32_t* flags, int32_t n, float* p, float* v) {
  if (!net || !flags || !p || !v) return fail(GA3C_EINVAL, "null argument");
  if (ticket < 0 || ticket >= (int32_t)net->lanes.size()) return fail(GA3C_EINVAL, "bad ticket %d", ticket);
  if (n < 1 || n > net->maxB) return fail(GA3C_EINVAL, "frames: %d requests outside [1,%d]", n, net->maxB);
  Lane* L = net->lanes[(size_t)ticket];
  if (!L->begun.exchange(false)) return fail(GA3C_ESTATE, "ticket %d: no batch was begun on that lane (or it was ended already)", ticket);
  const int64_t t0 = now_ns();
  const hipError_t he = hipEventSynchronize(L->done);      // left behind the batch by _begin
  int rc = he == hipSuccess ? GA3C_OK : fail(GA3C_EHIP, "hipEventSynchronize failed: %s", hipGetErrorString(he));
  const int want = L->frames_want;
  if (rc == GA3C_OK && want > 0) {
    note_predict_span(net, L);
    const int A = net->A;
    const float* hp = L->h_out;
    const float* hv = hp + (size_t)net->maxB * A;
    int k = 0;
    for (int i = 0; i < n; ++i) {
      if (flags[i] & 2u) continue;
      if (k >= want) { rc = fail(GA3C_EINVAL, "frames: the flags ask for more predictions than the batch was begun with (%d)", want); break; }
      memcpy(p + (size_t)i * A, hp + (size_t)k * A, (size_t)A * sizeof(float));
      v[i] = hv[k++];
    }
    stat_add(net, GA3C_STAT_PREDICT_CALLS, 1);
    stat_add(net, GA3C_STAT_PREDICT_ROWS, want);
    stat_add(net, GA3C_STAT_PREDICT_SYNC_NS, now_ns() - t0);
  }
  net->predict_inflight.fetch_sub(1, std::memory_order_relaxed);
  net->stream_busy[L->sidx].fetch_sub(1, std::memory_order_relaxed);
  L->mu.unlock();
  return rc;
}

int ga3c_net_serve_frames(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const uint32_t* flags, int32_t n,
                          float* p, float* v) {
  if (!p || !v) return fail(GA3C_EINVAL, "null argument");
  int32_t ticket = -1;
  CHK(ga3c_net_serve_frames_begin(net, offsets, agents, flags, n, &ticket));
  return ga3c_net_serve_frames_end(net, ticket, flags, n, p, v);
}

// rows named by (agent, plane sequence number) re-assembled from the plane history into train lane `t` (its mutex held)
static int stage_history_rows(ga3c_net* net, Stage& s, const int32_t* agents, const int64_t* seqs, const float* y_r,
                              const float* a, int32_t batch);

int ga3c_net_train_frames(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                          int32_t batch, float learning_rate, float beta, float* losses) {
  if (!net || !agents || !seqs || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  if (!net->fr.on || !net->fr.hist) return fail(GA3C_ESTATE, "frames: no plane history configured (ga3c_net_frames_config, history > 0)");
  return with_staged_batch(net, batch, [&](Stage& s) { return stage_history_rows(net, s, agents, seqs, y_r, a, batch); },
                           [&](TrainLane& t, Intake& in) { return train_enqueue(net, t, in, batch, learning_rate, beta); },
                           [&](TrainLane&, Intake& in) { return train_finish(net, in, losses); }, net->frames_in_line);
}

int ga3c_net_evaluate_frames(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                             int32_t batch, float beta, float* losses, float* d1, float* v, float* p) {
  if (!net || !agents || !seqs || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  if (!net->fr.on || !net->fr.hist) return fail(GA3C_ESTATE, "frames: no plane history configured (ga3c_net_frames_config, history > 0)");
  return with_staged_batch(net, batch, [&](Stage& s) { return stage_history_rows(net, s, agents, seqs, y_r, a, batch); },
                           [&](TrainLane& t, Intake&) { return evaluate_staged(net, t, batch, beta, losses, d1, v, p); },
      [](TrainLane&, Intake&) { return (int)GA3C_OK; });
}

// ---- state cache --------------------------------------------------------------------------------------------
int ga3c_net_state_cache_config(ga3c_net* net, int32_t max_agents, int32_t depth) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  if (max_agents < 1 || depth < 2 * CACHE_SLACK) return fail(GA3C_EINVAL, "state cache: %d agents x %d states", max_agents, depth);
  HIPCHK(hipSetDevice(net->cfg.device));
  CHK(sync_all(net));
  std::lock_guard<std::mutex> g(net->cache_mu);
  if (net->cache_ring) (void)hipFree(net->cache_ring);
  net->cache_ring = nullptr;
  net->cache_agents = net->cache_depth = 0;
  net->stat[GA3C_STAT_STATE_CACHE_BYTES].store(0, std::memory_order_relaxed);
  const size_t bytes = (size_t)max_agents * depth * XS;
  if (hipMalloc((void**)&net->cache_ring, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return fail(GA3C_EHIP, "state cache: cannot allocate %zu bytes (%d agents x %d states)", bytes, max_agents, depth);
  }
  net->cache_agents = max_agents;
  net->cache_depth = depth;
  net->cache_newest.assign((size_t)max_agents, -1);
  net->cache_tags.assign((size_t)max_agents * depth, -1);
  net->stat[GA3C_STAT_STATE_CACHE_BYTES].store((int64_t)bytes, std::memory_order_relaxed);
  return GA3C_OK;
}

// rows named (agent, request number) -> the intake's uint8 rows, HBM to HBM; returns / actions as for any staged batch
static int stage_cached_rows(ga3c_net* net, Stage& s, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                             int32_t batch) {
  if (batch < 1 || batch > net->maxB) return fail(GA3C_EINVAL, "batch %d outside [1,%d]", batch, net->maxB);
  CHK(cache_offsets(net, agents, seqs, batch, false, s.h_off));
  float* hy = s.h_in + (size_t)net->maxB * XS;
  float* ha = hy + net->maxB;
  SmallCopy sc{nullptr, nullptr, 0, nullptr, nullptr, 0};
  if (y_r) { memcpy(hy, y_r, (size_t)batch * sizeof(float)); sc.src0 = hy; sc.dst0 = s.yr; sc.n0 = batch; }
  if (a) { memcpy(ha, a, (size_t)batch * net->A * sizeof(float)); sc.src1 = ha; sc.dst1 = s.act; sc.n1 = batch * net->A; }
  RowOffsets ro;
  ro.n = 0;
  if (batch <= 192) { memcpy(ro.off, s.h_off, (size_t)batch * sizeof(int64_t)); ro.n = batch; }
  // (reads HBM, not the bus: a thread per 16 bytes, the rows are there in a few microseconds)
  hipLaunchKernelGGL(copy_rows_kernel<XS / 16>, dim3((XS / 16 + 255) / 256, batch), dim3(256), 0, s.st, net->cache_ring, s.h_off,
                     reinterpret_cast<uint4*>(s.xu8), batch, sc, ro);
  HIPCHK(hipGetLastError());
  s.x_u8 = true;
  return GA3C_OK;
}

int ga3c_net_train_cached(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                          int32_t batch, float learning_rate, float beta, float* losses) {
  if (!net || !agents || !seqs || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  return with_staged_batch(net, batch, [&](Stage& s) { return stage_cached_rows(net, s, agents, seqs, y_r, a, batch); },
                           [&](TrainLane& t, Intake& in) { return train_enqueue(net, t, in, batch, learning_rate, beta); },
                           [&](TrainLane&, Intake& in) { return train_finish(net, in, losses); }, true);
}

int ga3c_net_evaluate_cached(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                             int32_t batch, float beta, float* losses, float* d1, float* v, float* p) {
  if (!net || !agents || !seqs || !y_r || !a) return fail(GA3C_EINVAL, "null argument");
  return with_staged_batch(net, batch, [&](Stage& s) { return stage_cached_rows(net, s, agents, seqs, y_r, a, batch); },
                           [&](TrainLane& t, Intake&) { return evaluate_staged(net, t, batch, beta, losses, d1, v, p); },
                           [](TrainLane&, Intake&) { return (int)GA3C_OK; }, true);
}

int ga3c_net_frames_pushed(ga3c_net* net, int32_t agent, int64_t* pushed) {
  if (!net || !pushed) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (agent < 0 || agent >= f.maxA) return fail(GA3C_EINVAL, "frames: agent %d outside [0,%d)", agent, f.maxA);
  std::lock_guard<std::mutex> g(f.mu);
  *pushed = f.pushed[agent];
  return GA3C_OK;
}

static int stage_history_rows(ga3c_net* net, Stage& s, const int32_t* agents, const int64_t* seqs, const float* y_r,
                              const float* a, int32_t batch) {
  Frames& f = net->fr;
  {
    std::lock_guard<std::mutex> g(f.mu);
    for (int i = 0; i < batch; ++i) {
      if (agents[i] < 0 || agents[i] >= f.maxA) return fail(GA3C_EINVAL, "frames: agent %d outside [0,%d)", agents[i], f.maxA);
      const int64_t n = f.pushed[agents[i]];
      if (seqs[i] < CIN - 1 || seqs[i] >= n)
        return fail(GA3C_EINVAL, "row %d: agent %d has no state at plane %lld (%lld planes pushed)", i, agents[i], (long long)seqs[i], (long long)n);
      if (n - (seqs[i] - (CIN - 1)) > f.hist)
        return fail(GA3C_ESTATE, "row %d: plane %lld of agent %d has left the %d-plane history", i, (long long)(seqs[i] - (CIN - 1)), agents[i], f.hist);
    }
  }
  // the row descriptors ride in the lane's pinned offset array: seqs first, agent ids behind them
  int64_t* h_seq = s.h_off;
  int32_t* h_ag = reinterpret_cast<int32_t*>(s.h_in);
  for (int i = 0; i < batch; ++i) { h_seq[i] = seqs[i]; h_ag[i] = agents[i]; }
  const int64_t total = (int64_t)batch * (IMG * IMG / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 512) blocks = 512;      // (reads HBM, not the bus: two workgroups per CU leave the other wave slots to a step in flight)
  // returns and actions ride in the same launch (staged behind the x region of h_in: the ids at its start stay untouched)
  float* hy = s.h_in + (size_t)net->maxB * XS;
  float* ha = hy + net->maxB;
  SmallCopy sc{nullptr, nullptr, 0, nullptr, nullptr, 0};
  if (y_r) { memcpy(hy, y_r, (size_t)batch * sizeof(float)); sc.src0 = hy; sc.dst0 = s.yr; sc.n0 = batch; }
  if (a) { memcpy(ha, a, (size_t)batch * net->A * sizeof(float)); sc.src1 = ha; sc.dst1 = s.act; sc.n1 = batch * net->A; }
  HistRows hr;
  hr.n = 0;
  if (batch <= 192) {
    for (int i = 0; i < batch; ++i) { hr.seq[i] = seqs[i]; hr.agent[i] = agents[i]; }
    hr.n = batch;
  }
  hipLaunchKernelGGL(gather_history_kernel, dim3(blocks), dim3(256), 0, s.st, f.ring, h_ag, h_seq, f.hist, IMG * IMG, s.xu8, batch, sc, hr);
  HIPCHK(hipGetLastError());
  s.x_u8 = true;
  return GA3C_OK;
}

int ga3c_net_frames_upload(ga3c_net* net, const uint8_t* rgb, int32_t n) {
  if (!net || !rgb) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (n < 1 || n > f.maxA) return fail(GA3C_EINVAL, "frames: %d frames outside [1,%d]", n, f.maxA);
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> g(f.mu);
  HIPCHK(hipMemcpy(f.d_rgb, rgb, (size_t)n * f.frame_bytes, hipMemcpyHostToDevice));
  return GA3C_OK;
}

int ga3c_net_time_frames(ga3c_net* net, int32_t n, int32_t iters, float* elapsed_ms) {
  if (!net || !elapsed_ms) return fail(GA3C_EINVAL, "null argument");
  Frames& f = net->fr;
  if (!f.on) return fail(GA3C_ESTATE, "frames: call ga3c_net_frames_config first");
  if (n < 1 || n > f.maxA || iters < 1) return fail(GA3C_EINVAL, "frames: bad n / iters");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> g(f.mu);
  for (int i = 0; i < n; ++i) { f.h_agents[i] = i; f.h_reset[i] = 0; }
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  CHK(launch_frames(net, f.d_rgb, f.h_agents, f.h_reset, nullptr, n));   // warm
  HIPCHK(hipEventRecord(e0, f.st));
  for (int it = 0; it < iters; ++it) CHK(launch_frames(net, f.d_rgb, f.h_agents, f.h_reset, nullptr, n));
  HIPCHK(hipEventRecord(e1, f.st));
  HIPCHK(hipEventSynchronize(e1));
  HIPCHK(hipEventElapsedTime(elapsed_ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int i = 0; i < n; ++i) f.filled[i] = CIN;   // iters + 1 >= ... pushes went into queues 0..n-1
  return GA3C_OK;
}

int ga3c_net_upload(ga3c_net* net, const float* x, const float* y_r, const float* a, int32_t batch) {
  if (!net || !x) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  ResidentHold hold(net->tr);
  net->resident_gen.fetch_add(1, std::memory_order_relaxed);
  CHK(stage_train_inputs(net, net->tr, x, false, y_r, a, batch));
  HIPCHK(hipStreamSynchronize(net->tr.st));
  return GA3C_OK;
}

int ga3c_net_upload_u8(ga3c_net* net, const uint8_t* x, const float* y_r, const float* a, int32_t batch) {
  if (!net || !x) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  ResidentHold hold(net->tr);
  net->resident_gen.fetch_add(1, std::memory_order_relaxed);
  CHK(stage_train_inputs(net, net->tr, x, true, y_r, a, batch));
  HIPCHK(hipStreamSynchronize(net->tr.st));
  return GA3C_OK;
}

static int resident_predict_locked(ga3c_net* net, int B) {
  int idx;
  {
    std::shared_lock<std::shared_mutex> lk(net->wmu);
    idx = net->latest;
  }
  return launch_forward(net, net->tr.f, idx, B, net->tr.st, false, nullptr, 0.f);
}

int ga3c_net_predict_resident(ga3c_net* net, int32_t batch) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  return resident_predict_locked(net, batch);
}

int ga3c_net_train_resident(ga3c_net* net, int32_t batch, float learning_rate, float beta) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  CHK(train_grads(net, net->tr, batch, beta, true, learning_rate));
  return train_apply(net, net->tr, learning_rate);
}

int ga3c_net_sync(ga3c_net* net) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  return sync_all(net);
}

int ga3c_net_time_resident(ga3c_net* net, int32_t mode, int32_t batch, int32_t iters, float learning_rate,
                           float beta, float* elapsed_ms) {
  if (!net || !elapsed_ms) return fail(GA3C_EINVAL, "null argument");
  if (iters < 1 || (mode != 0 && mode != 1)) return fail(GA3C_EINVAL, "bad mode/iters");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  TrainLane& t = net->tr;
  HIPCHK(hipEventRecord(t.ev0, t.st));
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) {
      CHK(resident_predict_locked(net, batch));
    } else {
      CHK(train_grads(net, net->tr, batch, beta, true, learning_rate));
      CHK(train_apply(net, net->tr, learning_rate));
    }
  }
  HIPCHK(hipEventRecord(t.ev1, t.st));
  HIPCHK(hipEventSynchronize(t.ev1));
  HIPCHK(hipEventElapsedTime(elapsed_ms, t.ev0, t.ev1));
  return GA3C_OK;
}

int ga3c_net_time_predict_lanes(ga3c_net* net, int32_t batch, int32_t iters, int32_t nlanes, float* elapsed_ms) {
  // `iters` resident prediction steps dealt round-robin to `nlanes` prediction lanes (= NP predictor threads, each
  // with its own stream and workspace); every lane holds a copy of the batch uploaded with ga3c_net_upload.
  if (!net || !elapsed_ms) return fail(GA3C_EINVAL, "null argument");
  if (iters < 1 || nlanes < 1 || nlanes > (int)net->lanes.size()) return fail(GA3C_EINVAL, "bad iters/nlanes");
  if (batch < 1 || batch > net->maxB) return fail(GA3C_EINVAL, "bad batch");
  HIPCHK(hipSetDevice(net->cfg.device));
  ResidentHold hold(net->tr);
  CHK(sync_all(net));
  // every lane works on its own copy of the resident batch (as every ThreadPredictor has its own staging): made once per
  // uploaded batch, not per call -- the K steps of a timed block then start from inputs that are already in place
  const uint64_t gen = net->resident_gen.load(std::memory_order_relaxed);
  for (int l = 0; l < nlanes; ++l) {
    Lane* L = net->lanes[l];
    Fwd& lf = L->f;
    if (L->staged_gen == gen && L->staged_rows >= batch && L->staged_u8 == net->tr.f.x_u8) continue;
    lf.x_u8 = net->tr.f.x_u8;
    if (lf.x_u8) HIPCHK(hipMemcpy(lf.xu8, net->tr.f.xu8, (size_t)batch * XS, hipMemcpyDeviceToDevice));
    else HIPCHK(hipMemcpy(lf.x, net->tr.f.x, (size_t)batch * XS * sizeof(float), hipMemcpyDeviceToDevice));
    L->staged_gen = gen; L->staged_rows = batch; L->staged_u8 = lf.x_u8;
  }
  int idx;
  {
    std::shared_lock<std::shared_mutex> lk(net->wmu);
    idx = net->latest;   // == cur after the sync_all above
  }
  // one host thread per lane, as the engine's NP predictor threads drive them (ThreadPredictor.py:45-66); lane l takes the
  // steps l, l + nlanes, ... of the `iters`.  The threads are persistent (LaneDrivers): the clock brackets exactly the
  // launches and the lanes' stream synchronisation, not thread creation.
  if (nlanes == 1) {
    Lane* L = net->lanes[0];
    PredictInFlight inflight(net);
    const auto h0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventRecord(L->tm0, L->st));
    for (int i = 0; i < iters; ++i) CHK(lane_step(net, *L, idx, batch, STEP_RESIDENT, nullptr, nullptr));
    HIPCHK(hipEventRecord(L->tm1, L->st));
    HIPCHK(hipEventSynchronize(L->tm1));
    *elapsed_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    HIPCHK(hipEventElapsedTime(&net->lanes_gpu_ms, L->tm0, L->tm1));
    return GA3C_OK;
  }
  LaneDrivers& d = net->drv;
  while ((int)d.th.size() < nlanes) {
    const int l = (int)d.th.size();
    d.th.emplace_back(lane_driver_main, net, l);
  }
  uint64_t seq;
  {
    std::lock_guard<std::mutex> lk(d.mu);
    d.batch = batch; d.iters = iters; d.nlanes = nlanes; d.idx = idx;
    d.rcs.assign((size_t)nlanes, GA3C_OK);
    d.errs.assign((size_t)nlanes, std::string());
    d.ready.store(0);
    d.done.store(0);
    seq = ++d.seq;
    d.seq_hint.store(seq, std::memory_order_release);
  }
  d.cv.notify_all();
  while (d.ready.load(std::memory_order_acquire) < nlanes) __builtin_ia32_pause();   // every driver is awake and spinning
  const auto h0 = std::chrono::steady_clock::now();
  d.go.store(seq, std::memory_order_release);
  for (unsigned spin = 0; d.done.load(std::memory_order_acquire) < nlanes; ++spin)
    if (spin > 4096) sched_yield(); else __builtin_ia32_pause();
  *elapsed_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
  for (int l = 0; l < nlanes; ++l)
    if (d.rcs[l] != GA3C_OK) return fail(d.rcs[l], "prediction lane %d failed: %s", l, d.errs[l].c_str());
  net->lanes_gpu_ms = -1.f;                                  // worked out by ga3c_net_last_lanes_gpu_ms, outside the caller's clock
  net->lanes_gpu_n = nlanes;
  return GA3C_OK;
}

int ga3c_net_last_lanes_gpu_ms(ga3c_net* net, float* gpu_ms) {
  if (!net || !gpu_ms) return fail(GA3C_EINVAL, "null argument");
  if (net->lanes_gpu_ms < 0.f) {
    // the block as the GPU saw it: from the earliest lane's start event to the latest lane's end event
    HIPCHK(hipSetDevice(net->cfg.device));
    float span = 0.f;
    for (int i = 0; i < net->lanes_gpu_n; ++i)
      for (int j = 0; j < net->lanes_gpu_n; ++j) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, net->lanes[i]->tm0, net->lanes[j]->tm1));
        if (ms > span) span = ms;
      }
    net->lanes_gpu_ms = span;
  }
  *gpu_ms = net->lanes_gpu_ms;
  return GA3C_OK;
}

int ga3c_net_time_train_lanes(ga3c_net* net, int32_t batch, int32_t iters, int32_t nlanes, float learning_rate,
                              float beta, float* elapsed_ms) {
  // `iters` resident train steps dealt round-robin to `nlanes` train lanes of a Hogwild net (the NT trainer threads of
  // Config.TRAINERS); every lane gets a copy of the batch uploaded with ga3c_net_upload[_u8].
  if (!net || !elapsed_ms) return fail(GA3C_EINVAL, "null argument");
  if (iters < 1 || nlanes < 1 || nlanes > (int)net->xtr.size() + 1) return fail(GA3C_EINVAL, "bad iters/nlanes");
  if (nlanes > 1 && !net->hogwild) return fail(GA3C_ESTATE, "net was not created with train_lanes >= 2");
  if (batch < 1 || batch > net->maxB) return fail(GA3C_EINVAL, "bad batch");
  HIPCHK(hipSetDevice(net->cfg.device));
  ResidentHold hold(net->tr);
  CHK(sync_all(net));
  TrainLane& t0 = net->tr;
  for (int l = 1; l < nlanes; ++l) {
    TrainLane& t = *net->xtr[l - 1];
    t.f.x_u8 = t0.f.x_u8;
    if (t.f.x_u8) HIPCHK(hipMemcpy(t.f.xu8, t0.f.xu8, (size_t)batch * XS, hipMemcpyDeviceToDevice));
    else HIPCHK(hipMemcpy(t.f.x, t0.f.x, (size_t)batch * XS * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(t.yr, t0.yr, (size_t)batch * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(t.act, t0.act, (size_t)batch * net->A * sizeof(float), hipMemcpyDeviceToDevice));
  }
  const auto h0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) {
    TrainLane& t = (i % nlanes) == 0 ? net->tr : *net->xtr[(i % nlanes) - 1];
    CHK(train_grads(net, t, batch, beta, true, learning_rate));
    CHK(train_apply(net, t, learning_rate));
  }
  CHK(sync_all(net));
  *elapsed_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
  return GA3C_OK;
}

int ga3c_net_time_kernel(ga3c_net* net, const char* kernel, int32_t batch, int32_t iters, float* elapsed_ms) {
  // Each launch carries its own start/stop events (hipExtLaunchKernelGGL), so the sum is pure kernel
  // execution time on the train lane's stream, without launch gaps.  Buffers hold whatever the last
  // step left there; "rmsprop" runs with lr = 0 but does advance the `ms` slot (use a scratch net).
  if (!net || !kernel || !elapsed_ms) return fail(GA3C_EINVAL, "null argument");
  if (iters < 1 || batch < 1 || batch > net->maxB) return fail(GA3C_EINVAL, "bad batch/iters");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  TrainLane& t = net->tr;
  const float* th = net->theta[net->latest];
  float* g = net->grad;
  const int B = batch;
  const std::string k(kernel);
  HIPCHK(hipStreamSynchronize(t.st));
  double total = 0.0;
#define TL(kern, grid, ...) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, t.st, t.ev0, t.ev1, 0, __VA_ARGS__)
  for (int i = 0; i < iters; ++i) {
    if (k == "conv1_fwd") {
      TL(conv1_fwd_kernel<false>, dim3(B * 7), (const void*)t.f.x, th + OFF_W1, th + OFF_B1, t.f.n1, B);
    } else if (k == "conv1_fwd_u8") {
      TL(conv1_fwd_kernel<true>, dim3(B * 7), (const void*)t.f.xu8, th + OFF_W1, th + OFF_B1, t.f.n1, B);
    } else if (k == "conv1_dw_u8") {
      TL(conv1_dw_kernel<true>, dim3(conv1_dw_blocks(net, B)), (const void*)t.f.xu8, t.dn1, t.slab1, B * 7);
    } else if (k == "conv2_fwd") {
      TL(conv2_fwd_kernel<false>, dim3(B * 2), t.f.n1, th + OFF_W2, th + OFF_B2, t.f.n2, B);
    } else if (k == "conv2_fwd_q") {
      TL(conv2_fwd_kernel<true>, dim3(B * 4), t.f.n1, th + OFF_W2, th + OFF_B2, t.f.n2, B);
    } else if (k == "conv_stack_fwd") {
      hipExtLaunchKernelGGL((conv_stack_fwd_kernel<false, false>), dim3(B * 2), dim3(1024), CS_LDS_FLOATS * sizeof(float), t.st,
                            t.ev0, t.ev1, 0, (const void*)t.f.x, net->theta_pk[net->latest] + PK_W1F, th + OFF_B1, net->theta_pk[net->latest] + PK_W2F, th + OFF_B2, t.f.n1, t.f.n2, B,
                            (const int64_t*)nullptr, SrcOffsets{}, (uint8_t*)nullptr);
    } else if (k == "conv_stack_fwd_train") {
      hipExtLaunchKernelGGL((conv_stack_fwd_kernel<true, false>), dim3(B * 2), dim3(1024), CS_LDS_FLOATS * sizeof(float), t.st,
                            t.ev0, t.ev1, 0, (const void*)t.f.x, net->theta_pk[net->latest] + PK_W1F, th + OFF_B1, net->theta_pk[net->latest] + PK_W2F, th + OFF_B2, t.f.n1, t.f.n2, B,
                            (const int64_t*)nullptr, SrcOffsets{}, (uint8_t*)nullptr);
    } else if (k == "conv_stack_fwd_u8") {
      hipExtLaunchKernelGGL((conv_stack_fwd_kernel<false, true>), dim3(B * 2), dim3(1024), CS_LDS_FLOATS * sizeof(float), t.st,
                            t.ev0, t.ev1, 0, (const void*)t.f.xu8, net->theta_pk[net->latest] + PK_W1F, th + OFF_B1, net->theta_pk[net->latest] + PK_W2F, th + OFF_B2, t.f.n1, t.f.n2, B,
                            (const int64_t*)nullptr, SrcOffsets{}, (uint8_t*)nullptr);
    } else if (k == "dense1_fwd" || k == "dense1_fwd_frag") {
      const bool keep = net->d1f_tile;
      if (k == "dense1_fwd_frag") net->d1f_tile = false;
      const int rc = launch_dense1_fwd(net, t.f.n2, net->theta_pk[net->latest], t.f.part, B, dense_ks(B), t.st, t.ev0, t.ev1);
      net->d1f_tile = keep;
      CHK(rc);
    } else if (k == "conv1_dw") {
      TL(conv1_dw_kernel<false>, dim3(conv1_dw_blocks(net, B)), (const void*)t.f.x, t.dn1, t.slab1, B * 7);
    } else if (k == "conv2_dw") {
      TL(conv2_dw_kernel<2>, dim3(B < 256 ? B : 256, 4), t.f.n1, t.dn2, t.slab2, B);
    } else if (k == "conv_dw_pair") {
      const int nch1 = conv1_dw_blocks(net, B), nch2 = B < 256 ? B : 256;
      TL(conv_dw_pair_kernel<false>, dim3(nch1 + 4 * nch2), (const void*)t.f.x, t.dn1, t.slab1, B * 7, nch1, t.f.n1, t.dn2, t.slab2, B, nch2);
    } else if (k == "conv2_dw_occ3") {
      TL(conv2_dw_kernel<3>, dim3(B < 256 ? B : 256, 4), t.f.n1, t.dn2, t.slab2, B);
    } else if (k == "conv2_dx") {
      hipExtLaunchKernelGGL(conv2_dx_kernel, dim3(B, 2), dim3(512), 0, t.st, t.ev0, t.ev1, 0, t.dn2, net->theta_pk[net->latest] + PK_W2DX, t.f.n1, t.dn1, B);
    } else if (k == "dense1_dw") {
      HeadBwdArgs hb;
      hb.B = B; hb.A = net->A; hb.d1 = t.f.d1; hb.dz = t.dz; hb.dv = t.dv; hb.lossrow = t.lossrow;
      hb.g_wp = g + OFF_WP; hb.g_bp = g + off_bp(net->A); hb.g_wv = g + OFF_WV; hb.g_bv = g + OFF_BV; hb.losses = t.losses;
      TL(dense1_dw_kernel, dim3(FLAT / 32 + net->A + 2, 2), t.f.n2, t.dd1, g + OFF_WD, g + OFF_BD, B, hb);
    } else if (k == "dense1_dx") {
      TL(dense1_dx_kernel, dim3(FLAT / 32, ((B + 15) / 16 + 3) / 4), t.dd1, th + OFF_WD, t.f.n2, t.dn2, B);
    } else if (k == "dense1_bwd") {
      Dense1BwdArgs d;
      d.n2 = t.f.n2; d.dd1 = t.dd1; d.wd = th + OFF_WD; d.g_wd = g + OFF_WD; d.g_bd = g + OFF_BD; d.dn2 = t.dn2; d.B = B;
      d.hb.B = B; d.hb.A = net->A; d.hb.d1 = t.f.d1; d.hb.dz = t.dz; d.hb.dv = t.dv; d.hb.lossrow = t.lossrow;
      d.hb.g_wp = g + OFF_WP; d.hb.g_bp = g + off_bp(net->A); d.hb.g_wv = g + OFF_WV; d.hb.g_bv = g + OFF_BV; d.hb.losses = t.losses;
      d.dw_gx = FLAT / 32 + net->A + 2; d.dw_blocks = 2 * d.dw_gx; d.dx_gx = FLAT / 32;
      d.dx_mt = dense_dx_mt(B);
      TL(dense1_bwd_kernel, dim3(d.dw_blocks + d.dx_gx * (((B + 16 * d.dx_mt - 1) / (16 * d.dx_mt) + 3) / 4)), d);
    } else if (k == "conv_bwd") {
      if (B > 128) return fail(GA3C_EINVAL, "conv_bwd runs up to 128 rows");
      hipExtLaunchKernelGGL(conv_bwd_kernel<false>, dim3(2 * B), dim3(1024), CB_LDS_FLOATS * sizeof(float), t.st,
                            t.ev0, t.ev1, 0, (const void*)t.f.x, t.f.n1, t.dn2, net->theta_pk[net->latest] + PK_W2DX, t.dn1, t.slab2,
                            t.slab1, B, (const float*)nullptr, FusedUpd{});
    } else if (k == "conv_bwd_wdstep") {
      if (B > 128) return fail(GA3C_EINVAL, "conv_bwd runs up to 128 rows");
      FusedUpd fu{};
      const int l = net->latest;
      fu.tin = net->theta[l]; fu.tout = net->theta[l]; fu.ms = net->ms; fu.mom = net->mom; fu.pk = net->theta_pk[l];
      fu.lr = 0.f; fu.omr = 0.f; fu.mu = 0.f; fu.eps = net->cfg.rmsprop_epsilon; fu.on = 1; fu.defer_wd = 1;
      hipExtLaunchKernelGGL((conv_bwd_kernel<false, true>), dim3(2 * B), dim3(1024), CB_LDS_FLOATS * sizeof(float), t.st,
                            t.ev0, t.ev1, 0, (const void*)t.f.x, t.f.n1, t.dn2, net->theta_pk[l] + PK_W2DX, t.dn1, t.slab2,
                            t.slab1, B, (const float*)(g + OFF_WD), fu);
    } else if (k == "dense1_bwd_tile" || k == "dense1_bwd_tile_notail") {
      Dense1TileArgs d;
      d.n2 = t.f.n2; d.dd1 = t.dd1; d.wd = th + OFF_WD; d.g_wd = g + OFF_WD; d.g_bd = g + OFF_BD; d.dn2 = t.dn2; d.B = B;
      d.hb.B = B; d.hb.A = net->A; d.hb.d1 = t.f.d1; d.hb.dz = t.dz; d.hb.dv = t.dv; d.hb.lossrow = t.lossrow;
      d.hb.g_wp = g + OFF_WP; d.hb.g_bp = g + off_bp(net->A); d.hb.g_wv = g + OFF_WV; d.hb.g_bv = g + OFF_BV; d.hb.losses = t.losses;
      d.role_blocks = net->A + 2 < 14 ? net->A + 2 : 14;
      memset(&d.upd, 0, sizeof d.upd);
      d.tail_lds = k == "dense1_bwd_tile" && net->d1b_tail && B > D1B_ROWS && B <= D1B_ROWS + D1B_TAIL_ROWS;
      if (d.tail_lds)
        hipExtLaunchKernelGGL((dense1_bwd_tile_kernel<0, true>), dim3(D1B_TILES + d.role_blocks), dim3(1024), D1B_LDS_FLOATS_TAIL * sizeof(float),
                              t.st, t.ev0, t.ev1, 0, d);
      else
        hipExtLaunchKernelGGL(dense1_bwd_tile_kernel<0>, dim3(D1B_TILES + d.role_blocks), dim3(1024), D1B_LDS_FLOATS * sizeof(float), t.st,
                              t.ev0, t.ev1, 0, d);
    } else if (k == "heads") {
      HeadArgs h;
      memset(&h, 0, sizeof h);
      const int ks = dense_ks(B);
      h.part = t.f.part; h.ks = ks; h.B = B; h.A = net->A;
      h.bd = th + OFF_BD; h.wv = th + OFF_WV; h.bv = th + OFF_BV; h.wp = th + OFF_WP; h.bp = th + off_bp(net->A);
      h.d1 = t.f.d1; h.z = t.f.z; h.p = t.f.p; h.v = t.f.v;
      h.log_eps = net->cfg.log_epsilon; h.min_policy = net->cfg.min_policy;
#define HEADS_T(AM) hipExtLaunchKernelGGL((heads_kernel<false, AM>), dim3((B + HEADS_WAVES - 1) / HEADS_WAVES), dim3(64 * HEADS_WAVES), 0, t.st, t.ev0, t.ev1, 0, h)
      if (net->A <= 8) HEADS_T(8);
      else if (net->A <= 24) HEADS_T(24);
      else HEADS_T(64);
#undef HEADS_T
    } else if (k == "slab_reduce") {
      const int nch1 = conv1_dw_blocks(net, B), nch2 = B < 256 ? B : 256;
      SlabSet s1{t.slab1, nch1, SLAB1, 256 * 16, g + OFF_W1, g + OFF_B1, (SLAB1 + 63) / 64, OFF_W1, OFF_B1};
      SlabSet s2{t.slab2, nch2, SLAB2, 256 * 32, g + OFF_W2, g + OFF_B2, (SLAB2 + 63) / 64, OFF_W2, OFF_B2};
      FusedUpd noupd;
      memset(&noupd, 0, sizeof noupd);
      hipExtLaunchKernelGGL(slab_reduce_kernel<false>, dim3(s1.nblocks + s2.nblocks), dim3(1024), 0, t.st, t.ev0, t.ev1, 0, s1, s2, noupd);
    } else if (k == "rmsprop") {
      const int blocks = RMS_WD_BLOCKS + (int)((net->n - (int64_t)FLAT * HID + 255) / 256);
      TL((rmsprop_kernel<false, false>), dim3(blocks), th, net->theta[net->latest], net->ms, net->mom, t.grad, net->n,
         0.0f, 1.0f - net->cfg.rmsprop_decay, 0.0f, net->cfg.rmsprop_epsilon, net->tt, t.scales,
         net->theta_pk[net->latest]);
    } else {
      return fail(GA3C_EINVAL, "unknown kernel '%s'", kernel);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventSynchronize(t.ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, t.ev0, t.ev1));
    total += ms;
  }
#undef TL
  *elapsed_ms = (float)total;
  return GA3C_OK;
}

int ga3c_net_fetch(ga3c_net* net, const char* name, float* out, int64_t count) {
  if (!net || !name || !out) return fail(GA3C_EINVAL, "null argument");
  HIPCHK(hipSetDevice(net->cfg.device));
  ResidentHold hold(net->tr);
  TrainLane& t = net->tr;
  const int64_t B = net->maxB, A = net->A;
  struct Ent { const char* n; const float* p; int64_t cap; };
  const Ent ents[] = {{"n1", t.f.n1, B * N1S}, {"n2", t.f.n2, B * FLAT}, {"d1", t.f.d1, B * HID}, {"z", t.f.z, B * A},
                      {"p", t.f.p, B * A}, {"v", t.f.v, B}, {"dz", t.dz, B * A}, {"dv", t.dv, B},
                      {"dd1", t.dd1, B * HID}, {"dn2", t.dn2, B * FLAT}, {"dn1", t.dn1, B * N1S}, {"x", t.f.x, B * XS}};
  for (const Ent& e : ents) {
    if (strcmp(e.n, name) == 0) {
      if (count < 1 || count > e.cap) return fail(GA3C_EINVAL, "count %lld outside [1,%lld] for '%s'", (long long)count, (long long)e.cap, name);
      HIPCHK(hipStreamSynchronize(t.st));
      HIPCHK(hipMemcpy(out, e.p, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
      return GA3C_OK;
    }
  }
  return fail(GA3C_EINVAL, "unknown buffer '%s'", name);
}

int ga3c_host_alloc(void** ptr, int64_t bytes) {
  if (!ptr || bytes < 1) return fail(GA3C_EINVAL, "bad argument");
  HIPCHK(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
  return GA3C_OK;
}

int ga3c_host_free(void* ptr) {
  if (ptr) HIPCHK(hipHostFree(ptr));
  return GA3C_OK;
}

int ga3c_comm_make_id(uint8_t id[GA3C_COMM_ID_BYTES]) {
  if (!id) return fail(GA3C_EINVAL, "null argument");
  static_assert(sizeof(ncclUniqueId) <= GA3C_COMM_ID_BYTES, "ncclUniqueId larger than the ABI token");
  ncclUniqueId uid;
  NCCLCHK(ncclGetUniqueId(&uid));
  memset(id, 0, GA3C_COMM_ID_BYTES);
  memcpy(id, &uid, sizeof uid);
  return GA3C_OK;
}

int ga3c_net_comm_init(ga3c_net* net, const uint8_t id[GA3C_COMM_ID_BYTES], int32_t rank, int32_t world) {
  if (!net || !id) return fail(GA3C_EINVAL, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(GA3C_EINVAL, "rank %d / world %d invalid", rank, world);
  if (net->comm) return fail(GA3C_ESTATE, "communicator already attached");
  HIPCHK(hipSetDevice(net->cfg.device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  NCCLCHK(ncclCommInitRank(&net->comm, world, uid, rank));
  net->world = world;
  net->rank = rank;
  net->comm_overlap = !(getenv("GA3C_COMM_OVERLAP") && atoi(getenv("GA3C_COMM_OVERLAP")) == 0);
  net->head_on_train_stream = !(getenv("GA3C_COMM_HEAD_INLINE") && atoi(getenv("GA3C_COMM_HEAD_INLINE")) == 0);
  HIPCHK(stream_take(net->cfg.device, false, &net->cst));
  for (hipEvent_t* e : {&net->ev_tail_ready, &net->ev_head_ready, &net->ev_comm_done})
    HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  return GA3C_OK;
}

int ga3c_net_stats(ga3c_net* net, int64_t* out, int32_t n, int32_t reset) {
  if (!net || !out || n < 0) return fail(GA3C_EINVAL, "bad argument");
  for (int i = 0; i < n && i < GA3C_STAT_COUNT; ++i)      // (the cache's size is a gauge: a reset leaves it)
    out[i] = (reset && i != GA3C_STAT_STATE_CACHE_BYTES) ? net->stat[i].exchange(0, std::memory_order_relaxed)
                                                         : net->stat[i].load(std::memory_order_relaxed);
  for (int i = GA3C_STAT_COUNT; i < n; ++i) out[i] = 0;
  return GA3C_OK;
}

int ga3c_net_comm_info(ga3c_net* net, int32_t* ranks, int32_t* rank, int32_t* device) {
  if (!net || !ranks || !rank) return fail(GA3C_EINVAL, "null argument");
  *ranks = 0;
  *rank = -1;
  if (device) *device = -1;
  if (!net->comm) return GA3C_OK;
  int n = 0, r = -1, d = -1;
  NCCLCHK(ncclCommCount(net->comm, &n));
  NCCLCHK(ncclCommUserRank(net->comm, &r));
  NCCLCHK(ncclCommCuDevice(net->comm, &d));
  *ranks = n;
  *rank = r;
  if (device) *device = d;
  return GA3C_OK;
}

int ga3c_net_time_allreduce(ga3c_net* net, int32_t iters, float* elapsed_ms) {
  // `iters` back-to-back all-reduces (sum) of the gradient arena -- the exchange step of a data-parallel train step, alone
  // on the train stream -- between two HIP events recorded on that stream.  Collective: every rank calls it.
  if (!net || !elapsed_ms || iters < 1) return fail(GA3C_EINVAL, "bad argument");
  if (!net->comm) return fail(GA3C_ESTATE, "no communicator attached");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  TrainLane& t = net->tr;
  NCCLCHK(ncclAllReduce(net->grad, net->grad, (size_t)net->n, ncclFloat, ncclSum, net->comm, t.st));   // warm
  HIPCHK(hipEventRecord(t.ev0, t.st));
  for (int i = 0; i < iters; ++i)
    NCCLCHK(ncclAllReduce(net->grad, net->grad, (size_t)net->n, ncclFloat, ncclSum, net->comm, t.st));
  HIPCHK(hipEventRecord(t.ev1, t.st));
  HIPCHK(hipEventSynchronize(t.ev1));
  HIPCHK(hipEventElapsedTime(elapsed_ms, t.ev0, t.ev1));
  HIPCHK(hipMemsetAsync(net->grad, 0, (size_t)net->n * sizeof(float), t.st));   // the repeated sums may have overflowed
  HIPCHK(hipStreamSynchronize(t.st));
  return GA3C_OK;
}

int ga3c_net_allreduce_grads(ga3c_net* net) {
  if (!net) return fail(GA3C_EINVAL, "null argument");
  if (!net->comm) return fail(GA3C_ESTATE, "no communicator attached");
  HIPCHK(hipSetDevice(net->cfg.device));
  std::lock_guard<std::mutex> tl(net->tr.mu);
  NCCLCHK(ncclAllReduce(net->grad, net->grad, (size_t)net->n, ncclFloat, ncclSum, net->comm, net->tr.st));
  HIPCHK(hipStreamSynchronize(net->tr.st));
  return GA3C_OK;
}

}  // extern "C"
