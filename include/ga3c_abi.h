/* ga3c_abi.h -- C ABI of libga3c_hip.so, the MI355X (gfx950) NetworkVP engine.
 *
 * This is the drop-in boundary for the reference's `Network` object
 * (/root/reference/ga3c, TensorFlow session calls).  Every entry point names the
 * reference interface it replaces.  Plain pointers and sizes only; no C++ types,
 * no exceptions cross this boundary.  All functions return 0 on success and a
 * negative GA3C_E* code on failure; ga3c_last_error() then holds a message for
 * the calling thread.
 *
 * Threading: one ga3c_net may be called concurrently from any number of host
 * threads (the reference calls predict from NP predictor threads and train from
 * NT trainer threads on one object without locks, Server.py:123-134,141-153).
 * Predictions run on per-call lanes; train steps are serialised; a prediction
 * sees either the weights before or after a train step, never a mix.
 */
#ifndef GA3C_ABI_H
#define GA3C_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GA3C_OK 0
#define GA3C_EINVAL (-1)   /* bad argument / shape */
#define GA3C_EHIP (-2)     /* HIP runtime error */
#define GA3C_ERCCL (-3)    /* RCCL error */
#define GA3C_ESTATE (-4)   /* call not valid in this state */
#define GA3C_ELOST (-5)    /* a row named by (agent, request number) is no longer in the state cache: the batch was not trained */

#define GA3C_FLAG_LOG_SOFTMAX 1u   /* Config.USE_LOG_SOFTMAX branch, NetworkVP_discrate.py:64-71 */
#define GA3C_FLAG_GRAD_CLIP 2u     /* Config.USE_GRAD_CLIP, tf.clip_by_average_norm, :120-123 */

#define GA3C_STATE_FLOATS 28224    /* 84*84*4 (Config.py:90-92) */
#define GA3C_MAX_ACTIONS 64

typedef struct ga3c_net ga3c_net;

/* Replaces the constructor arguments + Config reads of Network.__init__
 * (NetworkVP.py:37-46) and of the optimizer block (NetworkVP_discrate.py:99-105). */
typedef struct ga3c_net_config {
  int32_t device;          /* HIP device ordinal (Config.DEVICE 'gpu:N') */
  int32_t num_actions;     /* A */
  int32_t max_batch;       /* capacity in rows of one predict / train call */
  uint32_t flags;          /* GA3C_FLAG_* */
  float rmsprop_decay;     /* Config.RMSPROP_DECAY   (0.99) */
  float rmsprop_momentum;  /* Config.RMSPROP_MOMENTUM (0.0) */
  float rmsprop_epsilon;   /* Config.RMSPROP_EPSILON (0.1) */
  float log_epsilon;       /* Config.LOG_EPSILON     (1e-6) */
  float min_policy;        /* Config.MIN_POLICY      (0.0) */
  float grad_clip_norm;    /* Config.GRAD_CLIP_NORM  (40.0), used with GA3C_FLAG_GRAD_CLIP */
  int32_t predict_lanes;   /* concurrent prediction lanes (>=1; 0 -> 2) */
  int32_t train_lanes;     /* 0/1: synchronous train steps on one lane (default).  >= 2: Hogwild -- that many train
                              lanes update the weights in place from their own streams, concurrently and unlocked, as
                              the reference's NT trainer threads do (Server.py:132-134); not combinable with RCCL */
} ga3c_net_config;

const char* ga3c_last_error(void);
int ga3c_device_count(int32_t* count);
/* PCI address of a device ("0000:23:00.0", NUL-terminated into out[len]).  The host side places its threads and agent
 * processes on the cores next to that device (ga3c_amd/Placement.py); the reference leaves placement to TensorFlow's
 * `DEVICE` string alone (Config.py:62, NetworkVP.py:50). */
int ga3c_device_pci_bus_id(int32_t device, char* out, int32_t len);

/* Network.__init__ / session teardown.  Weights are zero until ga3c_net_set_arena(which = 0). */
int ga3c_net_create(const ga3c_net_config* cfg, ga3c_net** out);
int ga3c_net_destroy(ga3c_net* net);

/* Parameter arena.  Flat f32 in TensorFlow variable order and layout:
 * conv11/w[8,8,4,16] conv11/b[16] conv12/w[4,4,16,32] conv12/b[32] dense1/w[3872,256]
 * dense1/b[256] logits_v/w[256,1] logits_v/b[1] logits_p/w[256,A] logits_p/b[A].
 * Replaces get_variable_value / tf.train.Saver (NetworkVP.py:62-64,267-288).
 * which: 0 = weights, 1 = RMSProp `ms` slot, 2 = RMSProp `mom` slot, 3 = last gradient. */
int ga3c_net_param_count(ga3c_net* net, int64_t* count);
int ga3c_net_get_arena(ga3c_net* net, int32_t which, float* out, int64_t count);
int ga3c_net_set_arena(ga3c_net* net, int32_t which, const float* in, int64_t count);
int ga3c_net_get_step(ga3c_net* net, int64_t* step);        /* get_global_step, NetworkVP.py:233-235 */
int ga3c_net_set_step(ga3c_net* net, int64_t step);
/* The same variables BY NAME, for binders that do not want to re-derive the table above: get_variables_names /
 * get_variable_value (NetworkVP.py:284-288) and the name-keyed tf.train.Saver (NetworkVP.py:62-64).
 *   ga3c_net_num_params          10
 *   ga3c_net_param_name(i)       "conv11/w", "conv11/b", "conv12/w", "conv12/b", "dense1/w", "dense1/b", "logits_v/w",
 *                                "logits_v/b", "logits_p/w", "logits_p/b" (i in arena order; NULL outside [0, 10))
 *   ga3c_net_param_info          offset and element count of the variable inside the arena, its rank and shape (<= 4 dims)
 *   ga3c_net_get_param/set_param one variable of arena `which` (0 weights, 1 `ms`, 2 `mom`, 3 last gradient; set: 0..2);
 *                                `count` must equal the variable's element count.  A name may carry TensorFlow's ":0". */
int32_t ga3c_net_num_params(ga3c_net* net);
const char* ga3c_net_param_name(ga3c_net* net, int32_t index);
int ga3c_net_param_info(ga3c_net* net, const char* name, int64_t* offset, int64_t* count, int32_t* ndim, int64_t shape[4]);
int ga3c_net_get_param(ga3c_net* net, const char* name, int32_t which, float* out, int64_t count);
int ga3c_net_set_param(ga3c_net* net, const char* name, int32_t which, const float* in, int64_t count);
/* save / load (NetworkVP.py:267-282): the whole training state -- every variable, its two RMSProp slots and `step` -- in ONE
 * file.  The format is an uncompressed .npz (numpy.savez / numpy.load): members "<name>:0", "<name>/RMSProp:0",
 * "<name>/RMSProp_1:0" with the variable's shape, and "step" (int64 scalar); a TensorFlow checkpoint cannot be written
 * without TensorFlow.  The file name convention checkpoints/<model>_%08d (NetworkVP.py:267-272) is the caller's.  load refuses
 * a file whose shapes do not match this network (another action count) and leaves the network untouched then. */
int ga3c_net_save(ga3c_net* net, const char* path);
int ga3c_net_load(ga3c_net* net, const char* path);

/* predict_p_and_v (NetworkVP.py:248-252): x f32[B,84,84,4] NHWC host buffer ->
 * p f32[B,A] (softmax_p), v f32[B] (logits_v); z f32[B,A] (logits_p) if not NULL. */
int ga3c_net_predict(ga3c_net* net, const float* x, int32_t batch, float* p, float* v, float* z);
/* Same, states shipped as the uint8 frames of Environment._preprocess before its
 * `/128 - 1` (Environment.py:59-60); they stay uint8 in HBM and the conv kernels convert while
 * reading them, bit-identically to the f32 path. */
int ga3c_net_predict_u8(ga3c_net* net, const uint8_t* x, int32_t batch, float* p, float* v, float* z);

/* train (NetworkVP.py:254-257 = sess.run(train_op)): forward, loss, backward,
 * (all-reduce when a communicator is attached), RMSProp, global_step += 1.
 * y_r f32[B], a f32[B,A] one-hot.  losses (may be NULL) receives
 * {cost_p_1_agg, cost_p_2_agg, cost_v} (NetworkVP_discrate.py:61,83-84) of this rank's rows. */
int ga3c_net_train(ga3c_net* net, const float* x, const float* y_r, const float* a, int32_t batch,
                   float learning_rate, float beta, float* losses);
/* Same, states as uint8 frames (converted on the GPU exactly like ga3c_net_predict_u8). */
int ga3c_net_train_u8(ga3c_net* net, const uint8_t* x, const float* y_r, const float* a, int32_t batch,
                      float learning_rate, float beta, float* losses);
/* log (NetworkVP.py:259-265 = sess.run(summary_op) on the batch it is given): forward + loss of `batch` rows on the
 * current weights, no backward pass, no update, global_step unchanged.  The states come from exactly one of x (f32 host
 * buffer), x_u8 (uint8 frames) or offsets (rows in the registered segment, offsets_u8 as in ga3c_net_train_gather).
 * losses[3] = {cost_p_1_agg, cost_p_2_agg, cost_v}; d1 f32[B,256], v f32[B], p f32[B,A] (each may be NULL) receive the
 * activations the reference histograms (NetworkVP_discrate.py:143-146: denselayer, logits_v, softmax_p). */
int ga3c_net_evaluate(ga3c_net* net, const float* x, const uint8_t* x_u8, const int64_t* offsets, int32_t offsets_u8,
                      const float* y_r, const float* a, int32_t batch, float beta, float* losses, float* d1, float* v,
                      float* p);
/* The two halves of train, for tests and for callers that own the exchange step:
 * gradients only (left in arena 3), then the optimizer step on arena 3. */
int ga3c_net_compute_grads(ga3c_net* net, const float* x, const float* y_r, const float* a,
                           int32_t batch, float beta, float* losses);
int ga3c_net_apply_grads(ga3c_net* net, float learning_rate);

/* Device-resident path (inputs already in HBM; what bench.py times).
 * upload stages a batch into the train lane (y_r / a may be NULL for predict-only use). */
int ga3c_net_upload(ga3c_net* net, const float* x, const float* y_r, const float* a, int32_t batch);
int ga3c_net_upload_u8(ga3c_net* net, const uint8_t* x, const float* y_r, const float* a, int32_t batch);  /* same, uint8 frames */
int ga3c_net_predict_resident(ga3c_net* net, int32_t batch);   /* async on the train lane's stream */
int ga3c_net_train_resident(ga3c_net* net, int32_t batch, float learning_rate, float beta);
int ga3c_net_sync(ga3c_net* net);
/* Runs `iters` back-to-back resident steps (mode 0 = predict, 1 = train) between two HIP
 * events recorded on the stream the kernels run on; returns the elapsed milliseconds. */
int ga3c_net_time_resident(ga3c_net* net, int32_t mode, int32_t batch, int32_t iters,
                           float learning_rate, float beta, float* elapsed_ms);
/* `iters` resident prediction steps dealt round-robin over `nlanes` prediction lanes (the NP predictor threads of
 * Config.PREDICTORS, each with its own HIP stream); host wall-clock from first launch to all lanes drained. */
int ga3c_net_time_predict_lanes(ga3c_net* net, int32_t batch, int32_t iters, int32_t nlanes, float* elapsed_ms);
/* The block ga3c_net_time_predict_lanes timed last, as the GPU saw it: from the earliest lane's start event to the latest
 * lane's end event (milliseconds).  Unlike the host clock it does not contain the launch latency of the first kernel and
 * the wake-up of the waiting host threads, which at the driver's K = 20 are a tenth of a block. */
int ga3c_net_last_lanes_gpu_ms(ga3c_net* net, float* gpu_ms);
/* `iters` resident train steps dealt round-robin over `nlanes` train lanes of a net created with train_lanes >= 2
 * (nlanes = 1 works on any net); host wall-clock from the first launch to all lanes drained. */
int ga3c_net_time_train_lanes(ga3c_net* net, int32_t batch, int32_t iters, int32_t nlanes, float learning_rate,
                              float beta, float* elapsed_ms);
/* Same bracket around ONE kernel of the step (name as in DESIGN.md, e.g. "conv1_fwd"). */
int ga3c_net_time_kernel(ga3c_net* net, const char* kernel, int32_t batch, int32_t iters,
                         float* elapsed_ms);

/* Activations / per-sample gradients of the last resident or train-lane step, for parity tests:
 * name in {"n1","n2","d1","z","p","v","dz","dv","dd1","dn2","dn1"}.  "dn1" is that of the last ga3c_net_compute_grads
 * (train steps of up to 128 rows consume it on chip and do not store it). */
int ga3c_net_fetch(ga3c_net* net, const char* name, float* out, int64_t count);

/* Zero-copy intake from the shared-memory transport (include/ga3c_host.h): register the whole segment
 * once (hipHostRegister), then a batch is described by one byte offset per row into that segment and the GPU
 * gathers the states itself (uint8 -> f32 fused) -- no host-side gather, no staging copy.  Offsets must be
 * 16-byte aligned; u8 != 0 means rows are 28,224 uint8 frames, else 28,224 f32.
 * Replaces the feed_dict copy of ThreadPredictor.py:57-58 / ThreadTrainer.py:52-59 + NetworkVP.py:252,257. */
int ga3c_net_register_host(ga3c_net* net, void* base, int64_t bytes);
int ga3c_net_unregister_host(ga3c_net* net);
int ga3c_net_predict_gather(ga3c_net* net, const int64_t* offsets, int32_t batch, int32_t u8, float* p, float* v,
                            float* z);
/* ga3c_net_predict_gather in two halves, for a predictor loop that answers the previous batch while the GPU works on this
 * one (ga3c_pq_serve_pipelined, include/ga3c_host.h): begin takes a lane, stages the offsets and ENQUEUES the step; end
 * waits for it, copies p[batch, A] and v[batch] out and gives the lane back.  Every begin must be followed by its end, from
 * the same thread (the lane stays taken in between); a thread may have two begun when the net has the lanes for it.  An end
 * for a ticket on which nothing is begun returns GA3C_ESTATE. */
int ga3c_net_predict_gather_begin(ga3c_net* net, const int64_t* offsets, int32_t batch, int32_t u8, int32_t* ticket);
int ga3c_net_predict_gather_end(ga3c_net* net, int32_t ticket, int32_t batch, float* p, float* v);
int ga3c_net_train_gather(ga3c_net* net, const int64_t* offsets, int32_t u8, const float* y_r, const float* a,
                          int32_t batch, float learning_rate, float beta, float* losses);

/* State cache (round 3): every state a trainer gathers out of the transport crossed PCIe once already, for its prediction --
 * and the gather's bursts on the bus are what training costs the predictions (profiles/README.md).  With a cache configured,
 * ga3c_net_predict_gather_begin_cached also keeps the uint8 state of every row it reads in HBM, in a ring of `depth`
 * states per agent: row i is named (agents[i], seqs[i]) -- the agent's id and the number of the request that carried the
 * state (ga3c_pq_request_seq, include/ga3c_host.h) -- and lands in slot seqs[i] % depth of its agent.  A train batch then
 * names its rows the same way (ProcessAgent.py:88-100 ships the state itself; NetworkVP.py:254-257 feeds it) and is
 * gathered HBM to HBM: ga3c_net_train_cached / ga3c_net_evaluate_cached = ga3c_net_train_gather / ga3c_net_evaluate on
 * those rows, bit for bit.  Every slot carries a tag, the request number it really holds, written once the step that stores
 * the state has been enqueued: a row whose slot holds another request (never stored, its step failed, or overwritten since)
 * or whose request is within 4 of falling out of its agent's window (a new prediction of that agent could overwrite it
 * before the copy has run) is refused with GA3C_ELOST and nothing is trained -- the caller drops the batch (ThreadTrainer
 * counts it).  `depth` is the caller's estimate of what an agent can have stored and not yet trained (Server.py: a multiple
 * of the agents' fair share of the rollouts in flight, not the worst case of one agent owning them all: 28,224 B per state,
 * 2.3 MB per agent at 80 states instead of 25 MB at 886); ga3c_net_stats reports the bytes held and the rows lost.  uint8
 * states, plain launches (no GA3C_GRAPHS: a replayed launch has its arguments baked in, the slots travel in them): up to 128
 * rows the conv stack stores the bytes it stages, beyond that the gathered batch is filed by a copy kernel behind the gather. */
int ga3c_net_state_cache_config(ga3c_net* net, int32_t max_agents, int32_t depth);
int ga3c_net_predict_gather_begin_cached(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const int64_t* seqs,
                                         int32_t batch, int32_t u8, int32_t* ticket);
int ga3c_net_train_cached(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                          int32_t batch, float learning_rate, float beta, float* losses);
int ga3c_net_evaluate_cached(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                             int32_t batch, float beta, float* losses, float* d1, float* v, float* p);

/* Frame front-end on the device: Environment._rgb2gray + _preprocess
 * (ga3c/Environment.py:52-60) and the 4-deep frame queue of :62-74, so that an actor ships only the emulator's raw
 * RGB frame and the [84,84,4] state never leaves HBM.  The arithmetic is the reference's, bit for bit (see
 * include/ga3c_host.h: ga3c_frame_preprocess, and oracle/frame_frontend.py): f64 gray, per-frame min/max bytescale,
 * Pillow's BILINEAR resize to 84x84.
 *   frames_config     height x width x channels (3 or 4) of the frames to come; one queue per agent in [0,max_agents);
 *                     history > 0 also keeps the last `history` planes of every agent in HBM for train_frames
 *   frames_preprocess stateless: n frames -> n uint8 planes [84*84]                 (= Environment._preprocess)
 *   frames_push       n frames into the queues of n DISTINCT agents; reset[i] != 0 clears that queue first
 *                     (Environment.reset, :86-90)                                    (= _update_frame_q);
 *                     seq_out[i] (may be NULL) = sequence number of the plane agent i just got, counting from 0
 *   frames_push_offsets  the same for frames lying in the registered transport segment, one byte offset each: the
 *                     predictor hands over what it popped, nothing is copied on the host
 *   train_frames      one training step whose row i is the state agent[i]'s queue held right after its plane seq[i]
 *                     was pushed (planes seq-3 .. seq), re-assembled from the plane history -- rollouts then carry
 *                     (agent, seq, return, action) instead of 28,224-byte states (ProcessAgent.py:175 / ThreadTrainer.py:49-59)
 *   frames_state      one agent's uint8 [84,84,4] state and queue depth; depth < 4 means "no state yet" (:64-65)
 *   predict_frames    forward pass on the queued states of `agents` (every queue must be full)
 * rgb may be pageable, pinned, device memory or lie in the registered transport segment (read in place then). */
int ga3c_net_frames_config(ga3c_net* net, int32_t max_agents, int32_t height, int32_t width, int32_t channels,
                           int32_t history);
int ga3c_net_frames_preprocess(ga3c_net* net, const uint8_t* rgb, int32_t n, uint8_t* planes);
int ga3c_net_frames_push(ga3c_net* net, const uint8_t* rgb, const int32_t* agents, const uint8_t* reset, int32_t n,
                         int64_t* seq_out);
int ga3c_net_frames_push_offsets(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const uint8_t* reset,
                                 int32_t n, int64_t* seq_out);
/* One predictor batch in raw-frame mode, one GPU round trip: the n popped requests' frames (byte offsets into the
 * registered segment) are pushed into their agents' queues -- flags[i] & 1: clear the queue first, flags[i] & 2: push
 * only (GA3C_REQ_RESET / GA3C_REQ_NO_PREDICT of include/ga3c_host.h) -- and rows i of p / v are filled for every request
 * that asked for a prediction.  This is the callback of the native predictor loop ga3c_pq_serve_frames. */
int ga3c_net_serve_frames(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const uint32_t* flags, int32_t n,
                          float* p, float* v);
/* The same batch in two halves, for a loop that answers batch k while the GPU works on batch k + 1 (ga3c_pq_serve_frames_pipelined;
 * ThreadPredictor.py:45-66 is one loop: predict, then scatter): _begin pushes the frames and enqueues the forward pass, and
 * keeps the prediction lane it took (*ticket names it); _end (same n and flags) waits for the batch and fills p / v as
 * ga3c_net_serve_frames does.  Every _begin must be followed by its _end; ga3c_net_serve_frames is the two back to back. */
int ga3c_net_serve_frames_begin(ga3c_net* net, const int64_t* offsets, const int32_t* agents, const uint32_t* flags, int32_t n,
                                int32_t* ticket);
int ga3c_net_serve_frames_end(ga3c_net* net, int32_t ticket, const uint32_t* flags, int32_t n, float* p, float* v);
int ga3c_net_train_frames(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                          int32_t batch, float learning_rate, float beta, float* losses);
/* ga3c_net_evaluate for rows named by (agent, plane sequence number), as ga3c_net_train_frames takes them. */
int ga3c_net_evaluate_frames(ga3c_net* net, const int32_t* agents, const int64_t* seqs, const float* y_r, const float* a,
                             int32_t batch, float beta, float* losses, float* d1, float* v, float* p);
/* Planes pushed into `agent`'s queue so far = sequence number its next plane gets.  Server.add_agent hands it to an
 * agent process that takes over the id of a removed one, so that its rollouts keep naming planes the way the device
 * counts them. */
int ga3c_net_frames_pushed(ga3c_net* net, int32_t agent, int64_t* pushed);
int ga3c_net_frames_state(ga3c_net* net, int32_t agent, uint8_t* state, int32_t* filled);
int ga3c_net_predict_frames(ga3c_net* net, const int32_t* agents, int32_t n, float* p, float* v, float* z);
/* bench helpers: n frames resident in HBM, then `iters` pushes of them timed with events on the frames stream */
int ga3c_net_frames_upload(ga3c_net* net, const uint8_t* rgb, int32_t n);
int ga3c_net_time_frames(ga3c_net* net, int32_t n, int32_t iters, float* elapsed_ms);

/* Pinned host memory for staging arrays (ThreadPredictor.py:46-47 `states`), so that
 * predict/train copy by DMA without an intermediate host copy. */
int ga3c_host_alloc(void** ptr, int64_t bytes);
int ga3c_host_free(void* ptr);

/* Where the calls of a running engine spend their time (no counterpart in the reference; tools/e2e_probe.py and the
 * engine legs of bench.py print the deltas).  out[i] for i < n, i < GA3C_STAT_COUNT; times in nanoseconds since the
 * net was created (or since the last call with reset != 0). */
enum {
  GA3C_STAT_PREDICT_CALLS = 0,     /* prediction calls (predict, predict_gather, predict_frames, serve_frames) */
  GA3C_STAT_PREDICT_ROWS,          /* rows they carried */
  GA3C_STAT_PREDICT_LANE_WAIT_NS,  /* waiting for a free prediction lane */
  GA3C_STAT_PREDICT_LAUNCH_NS,     /* host time to enqueue a step */
  GA3C_STAT_PREDICT_SYNC_NS,       /* from the last launch to the lane's stream being idle (GPU time + queueing) */
  GA3C_STAT_PREDICT_WEIGHT_WAITS,  /* steps that had to wait on the GPU for an optimizer step in flight (theta_ready) */
  GA3C_STAT_TRAIN_CALLS,           /* train calls */
  GA3C_STAT_TRAIN_ROWS,
  GA3C_STAT_TRAIN_STAGE_NS,        /* staging a batch into an intake: offsets, gather launch, y / a copies (host time) */
  GA3C_STAT_TRAIN_LANE_WAIT_NS,    /* waiting for the train lane (another thread's step) with the batch already staged */
  GA3C_STAT_TRAIN_LAUNCH_NS,       /* host time to enqueue forward + backward + update */
  GA3C_STAT_TRAIN_SYNC_NS,         /* from the last launch to the step's completion */
  GA3C_STAT_TRAIN_READER_WAITS,    /* cross-stream waits a step issued for prediction lanes still reading the buffer it overwrites */
  GA3C_STAT_PREDICT_GPU_NS,        /* GPU span of a prediction step, first kernel's start to last kernel's end; collected only with
                                      GA3C_TIME_PREDICTIONS=1 in the environment (two timing events per step) */
  GA3C_STAT_STATE_CACHE_BYTES,     /* gauge: bytes of HBM the state cache holds (0: none configured) */
  GA3C_STAT_STATE_CACHE_LOST,      /* rows of train / evaluate batches that were refused with GA3C_ELOST */
  GA3C_STAT_COUNT
};
int ga3c_net_stats(ga3c_net* net, int64_t* out, int32_t n, int32_t reset);

/* Data-parallel training over RCCL (no counterpart in the reference, which is
 * single-device: Config.py:62).  id is an opaque 128-byte token made on rank 0 and
 * handed to every rank by the launcher.  After comm_init, train and apply_grads
 * all-reduce (sum) the gradient arena across ranks before the optimizer step. */
#define GA3C_COMM_ID_BYTES 128
int ga3c_comm_make_id(uint8_t id[GA3C_COMM_ID_BYTES]);
int ga3c_net_comm_init(ga3c_net* net, const uint8_t id[GA3C_COMM_ID_BYTES], int32_t rank, int32_t world);
/* What the attached communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), not what the
 * launcher's environment claims: ranks = 0 and rank = -1 while no communicator is attached.  bench.py's `rccl_ranks`. */
int ga3c_net_comm_info(ga3c_net* net, int32_t* ranks, int32_t* rank, int32_t* device);
/* `iters` back-to-back all-reduces of the gradient arena between two HIP events on the train stream (bench.py's
 * allreduce_us).  Collective: every rank of the communicator calls it. */
int ga3c_net_time_allreduce(ga3c_net* net, int32_t iters, float* elapsed_ms);
int ga3c_net_allreduce_grads(ga3c_net* net);

#ifdef __cplusplus
}
#endif
#endif /* GA3C_ABI_H */
