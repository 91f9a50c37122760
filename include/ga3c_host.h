/* ga3c_host.h -- C ABI of libga3c_host.so: host-only pieces of the GA3C hot path
 * (no HIP; safe to load in forked agent processes).
 *
 *   returns     ProcessAgent._accumulate_rewards            (/root/reference/ga3c/ProcessAgent.py:69-84)
 *   transport   prediction_q / wait_q / training_q          (Server.py:73-75, ProcessAgent.py:64,102-107,175,
 *               of the reference, which are pickling            ThreadPredictor.py:45-66, ThreadTrainer.py:42-62)
 *               multiprocessing.Queues
 *
 * All functions return 0 (or a non-negative count / id where stated) on success and a negative
 * GA3C_H_E* code on failure.
 */
#ifndef GA3C_HOST_H
#define GA3C_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GA3C_H_OK 0
#define GA3C_H_EINVAL (-1)
#define GA3C_H_ESYS (-2)      /* shm_open / mmap / ftruncate failed (errno kept) */
#define GA3C_H_ETIMEOUT (-3)  /* a blocking call timed out (not an error for pollers) */
#define GA3C_H_ECLOSED (-4)   /* the transport was shut down */
#define GA3C_H_ECALLBACK (-5) /* ga3c_pq_serve: the predict callback failed (its own error string says why) */

const char* ga3c_host_last_error(void);

/* ---- returns --------------------------------------------------------------------------------
 * Bit-exact restatement of ProcessAgent._accumulate_rewards (ProcessAgent.py:69-84, call site
 * :148-149): IEEE f64, sequential `reward_sum = gamma * reward_sum`, all T rows kept, row T-1
 * keeps its raw reward; only (discounting && !use_intermediate_reward) writes anything back. */
int ga3c_returns_fork(const double* rewards, int32_t T, double gamma, double terminal_reward,
                      int32_t discounting, int32_t use_intermediate_reward, double* out);
/* Upstream-GA3C n-step return (the lines the fork commented out, ProcessAgent.py:83,146):
 * R = clip(r_t) + gamma R seeded with bootstrap_value; writes T-1 rows. */
int ga3c_returns_nstep(const double* rewards, int32_t T, double gamma, double bootstrap_value,
                       double rmin, double rmax, double* out);

/* ---- frame front-end (host side) -------------------------------------------------------------
 * Environment._rgb2gray + _preprocess (Environment.py:52-60) up to the uint8 plane the frame queue holds:
 *   gray  = f64 dot(rgb[..., :3], [.299, .587, .114])
 *   u8    = scipy.misc.bytescale(gray): stretch by the frame's own min / max to 0..255, +0.5, truncate
 *   plane = Pillow Image.resize((out_w, out_h), BILINEAR) of that 8-bit image
 * (the reference then maps k -> k/128 - 1 in f32; the transport keeps the byte).  rgb is [height][width][channels]
 * uint8 with channels >= 3; plane is [out_h][out_w].  Bit-exact with oracle/frame_frontend.py and with the HIP
 * implementation behind ga3c_net_frames_* (include/ga3c_abi.h). */
int ga3c_frame_preprocess(const uint8_t* rgb, int32_t height, int32_t width, int32_t channels, int32_t out_h,
                          int32_t out_w, uint8_t* plane);

/* ---- shared-memory transport ----------------------------------------------------------------
 * One POSIX shm segment holds
 *   agent slots   [max_agents] x { state (state_bytes), p[f32 x num_actions], v, request/response words }
 *   request ring  lock-free MPMC ring of agent ids (capacity >= max_agents: one request per agent in flight,
 *                 as wait_q = Queue(maxsize=1) in ProcessAgent.py:64)
 *   rollout slots [train_slots] x { rows, states[train_rows x state_bytes], returns f32[train_rows],
 *                 actions i32[train_rows] }, with a free ring and a ready ring (the bound of
 *                 Queue(maxsize=MAX_QUEUE_SIZE), Server.py:73)
 * The whole segment can be registered with HIP (hipHostRegister) so the GPU gathers states
 * straight from the slots.  Blocking uses futexes in the segment; nothing spins.            */
typedef struct ga3c_shm ga3c_shm;

typedef struct ga3c_shm_config {
  int32_t max_agents;
  int32_t num_actions;
  int32_t state_bytes;   /* 28224 (uint8 frames) or 112896 (f32 states) */
  int32_t train_slots;   /* rollouts in flight (MAX_QUEUE_SIZE) */
  int32_t train_rows;    /* rows per rollout slot (TIME_MAX + 1) */
  int32_t rollout_row_bytes; /* bytes per rollout row; 0 = state_bytes (rows carry whole states).  16 when rows only
                              * name a state kept on the device (agent id + plane sequence, ga3c_net_train_frames) */
  int32_t reserved[2];
} ga3c_shm_config;

int ga3c_shm_create(const char* name, const ga3c_shm_config* cfg, ga3c_shm** out);  /* server, before agents start */
int ga3c_shm_attach(const char* name, ga3c_shm** out);                              /* agent process */
int ga3c_shm_close(ga3c_shm* shm, int32_t unlink_segment);
int ga3c_shm_shutdown(ga3c_shm* shm);      /* wakes every waiter with GA3C_H_ECLOSED */
/* Removes the segment's NAME (owner only) and leaves every mapping alone: for a server that must end without unmapping --
 * the GPU may still be reading the registered segment -- and must not leave /dev/shm/ga3c_* behind. */
int ga3c_shm_unlink(ga3c_shm* shm);
/* Producers hold asynchronous signals back for the few instructions between taking a ring ticket and publishing the entry (an
 * agent process ended with SIGTERM -- Server.remove_agent, ProcessAgent.py has no counterpart: its queues are pipes -- must
 * not leave a ticket unfilled).  on = 0 switches that off for THIS process: for producers that are threads of one process,
 * whose signal masks hang on one kernel lock (the native agent threads of tests/native/native_agents.cpp); default 1. */
int ga3c_host_signal_hold(int32_t on);
void* ga3c_shm_base(ga3c_shm* shm);
int64_t ga3c_shm_bytes(ga3c_shm* shm);
int ga3c_shm_get_config(ga3c_shm* shm, ga3c_shm_config* cfg);
int64_t ga3c_shm_state_offset(ga3c_shm* shm, int32_t agent);      /* byte offset of an agent's state in the segment */
int64_t ga3c_shm_agent_stride(ga3c_shm* shm);
int64_t ga3c_shm_rollout_offset(ga3c_shm* shm, int32_t slot);     /* byte offset of a rollout slot's states */
int64_t ga3c_shm_rollout_stride(ga3c_shm* shm);

/* agent side of predict (ProcessAgent.py:102-107) */
void* ga3c_pq_state_ptr(ga3c_shm* shm, int32_t agent);
int ga3c_pq_submit(ga3c_shm* shm, int32_t agent);
/* The same with request flags for the predictor (device-side frame front-end: the slot then holds the emulator's raw
 * frame, Environment.py:76-93): RESET = the episode just started, clear the agent's frame queue first
 * (Environment.reset); NO_PREDICT = push the frame only, the queue is not full yet (ProcessAgent.py:127-129). */
#define GA3C_REQ_RESET 1u
#define GA3C_REQ_NO_PREDICT 2u
int ga3c_pq_submit_flags(ga3c_shm* shm, int32_t agent, uint32_t flags);
int ga3c_pq_request_flags(ga3c_shm* shm, const uint32_t* ids, int32_t n, uint32_t* flags);   /* predictor side */
int ga3c_pq_wait(ga3c_shm* shm, int32_t agent, float* p, float* v, int32_t timeout_ms);
/* One agent step's whole conversation with the predictor in ONE call (ProcessAgent.py:102-115: put, get, select_action):
 * `submit` != 0: copy `state` (state_bytes; NULL: the slot was filled in place) into the agent's slot and submit it with
 * `flags`; then wait up to timeout_ms for (p, v); then, with u >= 0, draw the action as ga3c_select_action(p, n, u) into
 * *action (u < 0: -1, the caller decides -- PLAY_MODE's argmax).  GA3C_H_ETIMEOUT leaves the request in flight: call again
 * with submit = 0 to go on waiting.  For an interpreted host this is one foreign call per step instead of four. */
int ga3c_pq_round_trip(ga3c_shm* shm, int32_t agent, const void* state, int32_t state_bytes, uint32_t flags, int32_t submit,
                       int32_t timeout_ms, double u, float* p, float* v, int32_t* action);
/* How long ga3c_pq_wait polls for the answer before the agent announces its sleep and waits on the slot's futex (0 = not at
 * all, the default: a GPU round trip is 60 us and more, and a polling agent holds a core).  An answer that arrives while the
 * agent is still awake costs neither side a system call: ga3c_pq_respond wakes only agents that have announced their sleep.
 * (wait_q.get() of ProcessAgent.py:105-106 blocks in a pipe read; this is the futex counterpart.) */
int ga3c_pq_set_spin(ga3c_shm* shm, int32_t spin_us);
/* What a wake costs on this machine: time from ga3c_pq_respond publishing an agent's answer to that agent returning from
 * ga3c_pq_wait, summed over all agents since the segment was created (each agent adds its own; read them when the agents are
 * quiet or take differences).  out6 = {answers, sum ns, max ns} of answers that were there already or came while the agent
 * polled, then the same three for answers the agent had gone to sleep for (FUTEX_WAIT): the second group's mean is the
 * scheduler's wake-to-run latency, which bounds an engine with many light agents.  No counterpart in
 * the reference (its wait is multiprocessing.Queue.get, ProcessAgent.py:98). */
int ga3c_pq_wake_latency(ga3c_shm* shm, int64_t* out6);
/* Environment._update_frame_q + _get_current_state (Environment.py:62-74) for the 4-deep frame queue kept as one
 * little-endian uint32 per pixel (byte c = frame c, oldest first): out[i] = (in[i] >> 8) | (plane[i] << 24), i < n.  The
 * words of `out` are the [84,84,4] uint8 state with the new plane as its newest frame; `in` is left untouched (experiences
 * of the running rollout still reference it).  The numpy expression took 6.8 us per agent step, this 0.5 us. */
int ga3c_frame_queue_push(const uint32_t* in, const uint8_t* plane, uint32_t* out, int32_t n);
/* ProcessAgent.select_action = np.random.choice(n, p=prediction) (ProcessAgent.py:109-115) given the uniform that call
 * would draw: float64 cumulative sum of the float32 probabilities, normalised by its last entry, first index whose
 * cumulative value exceeds u (searchsorted side='right'), clamped to n - 1.  Bit-for-bit the arithmetic of numpy's
 * RandomState.choice; the caller draws u with np.random.random_sample() so the global stream advances exactly as there. */
int32_t ga3c_select_action(const float* p, int32_t n, double u);
/* 1 when agent has no request in flight (every submit has been answered), 0 otherwise.  Server.add_agent (Server.py:106-110)
 * reuses the id of a removed agent only once this holds: an agent that was stopped while waiting may have left a request
 * behind, and a second submit on the same slot is refused while it is unanswered. */
int ga3c_pq_agent_idle(ga3c_shm* shm, int32_t agent);
/* predictor side (ThreadPredictor.py:50-55,61-63): block up to timeout for ONE request, then drain
 * without waiting up to max_ids; returns the count (0 on timeout). */
int ga3c_pq_pop_batch(ga3c_shm* shm, uint32_t* ids, int32_t max_ids, int32_t timeout_ms);
int ga3c_pq_respond(ga3c_shm* shm, const uint32_t* ids, int32_t n, const float* p, const float* v);
/* Optional linger for pop_batch (and the native loops built on it); 0 / 0 = off = the reference's greedy drain.  With
 * linger_us > 0 a predictor that holds fewer than min_batch requests after draining keeps polling for up to linger_us. */
int ga3c_pq_set_linger(ga3c_shm* shm, int32_t linger_us, int32_t min_batch);

/* The whole ThreadPredictor.run loop (ThreadPredictor.py:46-66) in native code, so that a predictor thread holds no
 * interpreter lock between batches: pop_batch -> byte offsets of the popped agents' states -> `predict` (the
 * signature of ga3c_net_predict_gather, include/ga3c_abi.h) -> respond.  Runs for about `slice_ms`, then returns
 * GA3C_H_OK so the caller can look at its exit flag and fold `stats` (accumulated, never reset here) into its
 * counters; GA3C_H_ECLOSED once the segment is shut down; GA3C_H_ECALLBACK if `predict` fails (the requests
 * of that batch are NOT answered: the caller is expected to stop the server). */
typedef int (*ga3c_predict_rows_fn)(void* net, const int64_t* offsets, int32_t batch, int32_t u8, float* p, float* v,
                                    float* z);
typedef struct ga3c_serve_stats {
  int64_t batches, served;
  int64_t ns_pop, ns_predict, ns_respond;
  int64_t largest_batch, reserved[2];
} ga3c_serve_stats;
int ga3c_pq_serve(ga3c_shm* shm, ga3c_predict_rows_fn predict, void* net, int32_t u8, int32_t max_batch,
                  int32_t slice_ms, ga3c_serve_stats* stats);
/* The same loop with the answering off the GPU's critical path: the loop answers batch k (a futex wake per agent that is
 * asleep, ~1 us each) after it has popped and ENQUEUED (`begin`) batch k+1, beside the GPU's work on it; with nothing
 * queued batch k is answered at once.  GA3C_RESPONDER in the environment moves the answering: 1 = to a helper thread of the
 * call, as soon as `end` has returned the results (round 3's default, worth it only while wakes land on cold cores);
 * 2 = loop and helper share a batch; 3 = the loop, always before it pops the next batch; 0 = the default above.  Nothing
 * is held when the call returns.  `begin` / `end` have the signatures of ga3c_net_predict_gather_begin / _end
 * (include/ga3c_abi.h).  Same return values. */
typedef int (*ga3c_predict_begin_fn)(void* net, const int64_t* offsets, int32_t batch, int32_t u8, int32_t* ticket);
typedef int (*ga3c_predict_end_fn)(void* net, int32_t ticket, int32_t batch, float* p, float* v);
int ga3c_pq_serve_pipelined(ga3c_shm* shm, ga3c_predict_begin_fn begin, ga3c_predict_end_fn end, void* net, int32_t u8,
                            int32_t max_batch, int32_t slice_ms, ga3c_serve_stats* stats);
/* ... and for an engine that keeps the states it reads (ga3c_net_state_cache_config, include/ga3c_abi.h): `begin` has the
 * signature of ga3c_net_predict_gather_begin_cached and also gets each row's name -- the agent's id and the number of the
 * request that carried the state.  ga3c_pq_request_seq: that number for the agent's newest request (in flight, or answered
 * last), which is how an agent names the state of an experience instead of shipping it (rollout rows of 16 bytes: request
 * number i64, agent id i32; ga3c_tq_collect's row_seq / row_agent). */
typedef int (*ga3c_predict_begin_cached_fn)(void* net, const int64_t* offsets, const int32_t* agents, const int64_t* seqs,
                                            int32_t batch, int32_t u8, int32_t* ticket);
int ga3c_pq_serve_pipelined_cached(ga3c_shm* shm, ga3c_predict_begin_cached_fn begin, ga3c_predict_end_fn end, void* net,
                                   int32_t u8, int32_t max_batch, int32_t slice_ms, ga3c_serve_stats* stats);
int ga3c_pq_request_seq(ga3c_shm* shm, int32_t agent, int64_t* seq);
/* The same loop for raw-frame requests (ga3c_pq_submit_flags): `serve` has the signature of ga3c_net_serve_frames
 * (include/ga3c_abi.h) and gets the popped slots' offsets, agent ids and request flags; stats->served counts the
 * predictions made (requests without GA3C_REQ_NO_PREDICT).  The loop answers a batch before it pops the next one
 * (GA3C_RESPONDER = 1 / 2: a helper thread of the call does, or shares it, as in ga3c_pq_serve_pipelined). */
typedef int (*ga3c_serve_frames_fn)(void* net, const int64_t* offsets, const int32_t* agents, const uint32_t* flags,
                                    int32_t n, float* p, float* v);
int ga3c_pq_serve_frames(ga3c_shm* shm, ga3c_serve_frames_fn serve, void* net, int32_t max_batch, int32_t slice_ms,
                         ga3c_serve_stats* stats);
/* ... and with the engine's call in two halves (ga3c_net_serve_frames_begin / _end): the loop pops batch k + 1, begins it,
 * answers batch k -- a system call per sleeping agent, ~1.2 us a row, beside the GPU's work on k + 1 instead of in front of
 * it -- and ends k + 1, but only when requests for at least a quarter of max_batch are already queued (GA3C_PIPELINE_MIN_QUEUED):
 * with fewer the held answers go out first, as in the loop above -- a small closed population of agents would otherwise
 * travel as twice as many, half as large batches.  With a batch waiting to be answered only requests that are already
 * queued are popped.  What the loop holds is answered before it returns; GA3C_RESPONDER = 3 answers every batch at once. */
typedef int (*ga3c_serve_frames_begin_fn)(void* net, const int64_t* offsets, const int32_t* agents, const uint32_t* flags,
                                          int32_t n, int32_t* ticket);
typedef int (*ga3c_serve_frames_end_fn)(void* net, int32_t ticket, const uint32_t* flags, int32_t n, float* p, float* v);
int ga3c_pq_serve_frames_pipelined(ga3c_shm* shm, ga3c_serve_frames_begin_fn begin, ga3c_serve_frames_end_fn end, void* net,
                                   int32_t max_batch, int32_t slice_ms, ga3c_serve_stats* stats);

/* training queue: agent side (ProcessAgent.py:175), trainer side (ThreadTrainer.py:49-59) */
int ga3c_tq_acquire(ga3c_shm* shm, int32_t timeout_ms);                 /* -> free slot id */
void* ga3c_tq_states(ga3c_shm* shm, int32_t slot);
float* ga3c_tq_returns(ga3c_shm* shm, int32_t slot);
int32_t* ga3c_tq_actions(ga3c_shm* shm, int32_t slot);
int ga3c_tq_commit(ga3c_shm* shm, int32_t slot, int32_t rows);
int ga3c_tq_pop(ga3c_shm* shm, int32_t timeout_ms);                     /* -> ready slot id */
int ga3c_tq_rows(ga3c_shm* shm, int32_t slot);
int ga3c_tq_release(ga3c_shm* shm, int32_t slot);
/* The batch assembly of ThreadTrainer.run (ThreadTrainer.py:48-59) in one call, for trainers that let the GPU read the rows out
 * of the slots: pop rollouts until the batch holds MORE than min_rows rows.  *rows / *n_slots are the state of the batch in
 * progress (0 / 0 to start one; kept across GA3C_H_ETIMEOUT so that the caller can look at its exit flag and call again).
 * Per rollout: its slot id -> slots[], the byte offsets of its rows in the segment -> row_offsets[], its returns and actions
 * -> returns[] / actions[] (row-aligned).  The slots stay the caller's until ga3c_tq_release(_many).  Waits timeout_ms for the
 * first rollout and hold_timeout_ms once slots are held.  GA3C_H_ESTARVED: slots are held, none is free and none is queued
 * -- the agents are blocked on the caller, which must give slots back before asking again. */
#define GA3C_H_ESTARVED 1
/* row_seq / row_agent (both or neither; 16-byte rollout rows only): the rows NAME states kept on the device -- plane sequence
 * number and agent id are decoded into these arrays and every slot is released at once (nothing is held, *n_slots stays 0). */
int ga3c_tq_collect(ga3c_shm* shm, int32_t min_rows, int32_t timeout_ms, int32_t hold_timeout_ms, int32_t* rows,
                    int32_t* n_slots, int32_t* slots, int64_t* row_offsets, float* returns, int32_t* actions,
                    int32_t cap_rows, int32_t cap_slots, int64_t* row_seq, int32_t* row_agent);
int ga3c_tq_release_many(ga3c_shm* shm, const int32_t* slots, int32_t n);
int ga3c_tq_ready_count(ga3c_shm* shm);
int ga3c_tq_free_count(ga3c_shm* shm);    /* slots no producer and no consumer holds; 0 = producers are blocked */

#ifdef __cplusplus
}
#endif
#endif /* GA3C_HOST_H */
