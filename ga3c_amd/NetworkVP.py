"""Network -- the reference's model object, backed by libga3c_hip.so instead of a TensorFlow session.

Same constructor and methods as the reference's discrete-action Network
(/root/reference/ga3c/NetworkVP.py:36-64,230-288 and NetworkVP_discrate.py:35-130), so Server,
ThreadPredictor and ThreadTrainer call it unchanged:
    Network(device, model_name, num_actions, state_dim)
    predict_p_and_v(x) -> [p, v]      train(x, y_r, a, x2, done, trainer_id)
    log(...)  save(episode)  load() -> episode   .learning_rate  .beta
The graph is the A3C conv->dense head (NetworkDNav.py:80-90 topology, NetworkVP.py:212-228 layers,
NetworkVP_discrate.py:58-85 heads/loss).  Every numeric step runs in the HIP library; this file only
marshals numpy buffers.  If the library is missing or no gfx950 device is present it raises.
"""
import ctypes as C
import glob
import os
import re
import threading

import numpy as np

from Config import Config
import _native as nat

PARAM_ORDER = ("conv11/w", "conv11/b", "conv12/w", "conv12/b", "dense1/w", "dense1/b",
               "logits_v/w", "logits_v/b", "logits_p/w", "logits_p/b")


def param_shapes(num_actions):
    return {"conv11/w": (8, 8, 4, 16), "conv11/b": (16,), "conv12/w": (4, 4, 16, 32), "conv12/b": (32,),
            "dense1/w": (3872, 256), "dense1/b": (256,), "logits_v/w": (256, 1), "logits_v/b": (1,),
            "logits_p/w": (256, num_actions), "logits_p/b": (num_actions,)}


def initial_arena(num_actions, seed):
    """U(-d, d), d = 1/sqrt(fan_in) (NetworkVP.py:214, NetworkDNav.py:258), flat in TF variable order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = param_shapes(num_actions)
    parts = []
    for name in PARAM_ORDER:
        base = name.split("/")[0]
        d = 1.0 / np.sqrt(np.prod(shapes[base + "/w"][:-1]))
        parts.append(rng.uniform(-d, d, size=shapes[name]).astype(np.float32).ravel())
    return np.concatenate(parts)


def _default_bucket_limits():
    """Bucket limits of TensorFlow's default histogram (tensorflow/core/lib/histogram/histogram.cc,
    InitDefaultBucketsInner: 1e-12 * 1.1^k up to 1e20, mirrored for negatives, 0 in between, DBL_MAX last) -- what
    tf.summary.histogram (NetworkVP_discrate.py:140-146) files its values under."""
    pos = []
    v = 1.0e-12
    while v < 1.0e20:
        pos.append(v)
        v *= 1.1
    pos.append(np.finfo(np.float64).max)
    return np.array([-x for x in reversed(pos)] + [0.0] + pos, dtype=np.float64)


_BUCKET_LIMITS = _default_bucket_limits()


def histogram_proto(values):
    """The fields of a TensorFlow HistogramProto for `values`: min, max, num, sum, sum_squares and the non-empty run of
    (bucket_limit, bucket) pairs; a value x is counted in the first bucket whose limit is > x (upper_bound)."""
    x = np.asarray(values, dtype=np.float64).ravel()
    idx = np.searchsorted(_BUCKET_LIMITS, x, side="right")
    counts = np.bincount(idx, minlength=_BUCKET_LIMITS.size)[:_BUCKET_LIMITS.size].astype(np.float64)
    nz = np.nonzero(counts)[0]
    lo, hi = (int(nz[0]), int(nz[-1]) + 1) if nz.size else (0, 0)
    return {"min": np.float64(x.min() if x.size else 0.0), "max": np.float64(x.max() if x.size else 0.0),
            "num": np.float64(x.size), "sum": np.float64(x.sum()), "sum_squares": np.float64(np.dot(x, x)),
            "bucket_limit": _BUCKET_LIMITS[lo:hi].copy(), "bucket": counts[lo:hi].copy()}


def _device_ordinal(device):
    m = re.search(r"(\d+)\s*$", str(device))
    return int(m.group(1)) if m else 0


class Network:
    def __init__(self, device, model_name, num_actions, state_dim, max_batch=None, predict_lanes=None, train_lanes=None):
        self.device = device
        self.model_name = model_name
        self.num_actions = int(num_actions)
        self.state_dim = state_dim
        self.learning_rate = Config.LEARNING_RATE_START
        self.beta = Config.BETA_START
        self.log_epsilon = Config.LOG_EPSILON
        if int(np.prod(state_dim)) != nat.STATE_FLOATS:
            raise ValueError("state_dim %r is not 84x84x4" % (state_dim,))
        if Config.DUAL_RMSPROP:
            raise ValueError("DUAL_RMSPROP is out of scope (SURVEY.md section 9, Q7)")
        if max_batch is None:
            max_batch = max(Config.PREDICTION_BATCH_SIZE,
                            Config.TRAIN_ROWS_MAX or (Config.TRAINING_MIN_BATCH_SIZE + Config.TIME_MAX + 1))
        self.max_batch = int(max_batch)
        self._lib = nat.hip_lib()
        cfg = nat.NetConfig()
        cfg.device = _device_ordinal(device)
        cfg.num_actions = self.num_actions
        cfg.max_batch = self.max_batch
        cfg.flags = (nat.FLAG_LOG_SOFTMAX if Config.USE_LOG_SOFTMAX else 0) | \
                    (nat.FLAG_GRAD_CLIP if Config.USE_GRAD_CLIP else 0)
        cfg.rmsprop_decay = Config.RMSPROP_DECAY
        cfg.rmsprop_momentum = Config.RMSPROP_MOMENTUM
        cfg.rmsprop_epsilon = Config.RMSPROP_EPSILON
        cfg.log_epsilon = Config.LOG_EPSILON
        cfg.min_policy = Config.MIN_POLICY
        cfg.grad_clip_norm = Config.GRAD_CLIP_NORM
        # a lane = workspace + pinned staging of one prediction in flight (the lanes share two HIP streams: ga3c_net_create);
        # predictor threads beyond the lanes wait for one.  One per configured predictor; with the dynamic adjustment on, NP
        # walks (ThreadDynamicAdjustment.py:95-144), so there are at least four -- spare lanes cost nothing since round 3
        # (round 2: a stream per lane, and the fifth stream landed on a lane's hardware queue).  GA3C_PREDICT_LANES overrides
        default_lanes = max(1, Config.PREDICTORS, 4 if Config.DYNAMIC_SETTINGS else 1)
        cfg.predict_lanes = int(predict_lanes or os.environ.get("GA3C_PREDICT_LANES") or default_lanes)
        if train_lanes is None:
            train_lanes = max(1, Config.TRAINERS) if Config.HOGWILD else 1
        cfg.train_lanes = int(train_lanes)
        handle = C.c_void_p()
        nat.check(self._lib.ga3c_net_create(C.byref(cfg), C.byref(handle)), "ga3c_net_create")
        self._h = handle
        n = C.c_int64()
        nat.check(self._lib.ga3c_net_param_count(self._h, C.byref(n)))
        self.param_count = n.value
        self._offsets = {}
        off = 0
        for name in PARAM_ORDER:
            size = int(np.prod(param_shapes(self.num_actions)[name]))
            self._offsets[name] = (off, size)
            off += size
        assert off == self.param_count
        self.set_arena(0, initial_arena(self.num_actions, Config.RANDOM_SEED))
        self._pinned = []
        self._log_lock = threading.Lock()
        self.last_losses = None

    def pinned_array(self, shape, dtype=np.float32):
        """Host staging array in HIP-pinned memory (freed by close()); predict/train DMA straight from it."""
        arr, p = nat.pinned_array(shape, dtype)
        self._pinned.append(p)
        return arr

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ga3c_net_destroy(self._h)
            self._h = None
            for p in self._pinned:
                nat.free_pinned(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- arenas -----------------------------------------------------------------------------
    def get_arena(self, which):
        out = np.empty(self.param_count, dtype=np.float32)
        nat.check(self._lib.ga3c_net_get_arena(self._h, which, nat.ptr(out), out.size), "ga3c_net_get_arena")
        return out

    def set_arena(self, which, flat):
        flat = nat.as_f32(flat).ravel()
        nat.check(self._lib.ga3c_net_set_arena(self._h, which, nat.ptr(flat), flat.size), "ga3c_net_set_arena")

    def get_global_step(self):
        s = C.c_int64()
        nat.check(self._lib.ga3c_net_get_step(self._h, C.byref(s)))
        return s.value

    def get_variables_names(self):
        """NetworkVP.py:284-285; the names come out of the library's own table (ga3c_net_param_name)."""
        n = self._lib.ga3c_net_num_params(self._h)
        return [self._lib.ga3c_net_param_name(self._h, i).decode() + ":0" for i in range(n)]

    def _param_info(self, name):
        off, count, ndim = C.c_int64(), C.c_int64(), C.c_int32()
        shape = (C.c_int64 * 4)()
        nat.check(self._lib.ga3c_net_param_info(self._h, name.encode(), C.byref(off), C.byref(count), C.byref(ndim), shape),
                  "ga3c_net_param_info")
        return off.value, count.value, tuple(shape[d] for d in range(ndim.value))

    def get_variable_value(self, name, which=0):
        """NetworkVP.py:287-288 (which = 1 / 2: the variable's RMSProp slots, 3: its last gradient)."""
        _, count, shape = self._param_info(name)
        out = np.empty(count, dtype=np.float32)
        nat.check(self._lib.ga3c_net_get_param(self._h, name.encode(), which, nat.ptr(out), count), "ga3c_net_get_param")
        return out.reshape(shape)

    def set_variable_value(self, name, value, which=0):
        flat = nat.as_f32(value).ravel()
        nat.check(self._lib.ga3c_net_set_param(self._h, name.encode(), which, nat.ptr(flat), flat.size), "ga3c_net_set_param")

    # ---- inference ---------------------------------------------------------------------------
    def _predict(self, x, want_z=False):
        b = int(x.shape[0])
        p = np.empty((b, self.num_actions), dtype=np.float32)
        v = np.empty((b,), dtype=np.float32)
        z = np.empty((b, self.num_actions), dtype=np.float32) if want_z else None
        zp = nat.ptr(z) if want_z else None
        if x.dtype == np.uint8:
            x = np.ascontiguousarray(x)
            nat.check(self._lib.ga3c_net_predict_u8(self._h, nat.ptr(x, nat.u8p), b, nat.ptr(p), nat.ptr(v), zp),
                      "ga3c_net_predict_u8")
        else:
            x = nat.as_f32(x)
            nat.check(self._lib.ga3c_net_predict(self._h, nat.ptr(x), b, nat.ptr(p), nat.ptr(v), zp),
                      "ga3c_net_predict")
        return p, v, z

    def predict_p_and_v(self, x):
        p, v, _ = self._predict(x)
        return [p, v]

    def predict_p_v_logits(self, x):
        return self._predict(x, want_z=True)

    def predict_p(self, x):
        return self._predict(x)[0]

    def predict_v(self, x):
        return self._predict(x)[1]

    def predict_single(self, x):
        return self.predict_p(x[None, :])[0]

    # ---- zero-copy intake from the shared-memory transport --------------------------------------
    def register_transport(self, transport):
        """Pin the transport's segment for the GPU; afterwards predict_slots / train_rows gather from it."""
        nat.check(self._lib.ga3c_net_register_host(self._h, C.c_void_p(transport.base), transport.nbytes),
                  "ga3c_net_register_host")
        self._transport_u8 = transport.state_bytes == nat.STATE_FLOATS

    def unregister_transport(self):
        """Unpin the segment (call before the transport is unmapped)."""
        nat.check(self._lib.ga3c_net_unregister_host(self._h), "ga3c_net_unregister_host")

    # ---- frame front-end on the device (Environment.py:52-74; include/ga3c_abi.h: ga3c_net_frames_*) ------------
    def frames_config(self, max_agents, height=210, width=160, channels=3, history=0):
        """history > 0: the device also keeps each agent's last `history` planes, from which train_frames re-assembles
        training rows."""
        nat.check(self._lib.ga3c_net_frames_config(self._h, max_agents, height, width, channels, history),
                  "ga3c_net_frames_config")
        self._frame_shape = (height, width, channels)

    def _frames_arg(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim == 3:
            rgb = rgb[None]
        if rgb.shape[1:] != self._frame_shape:
            raise ValueError("frames of shape %s, configured for %s" % (rgb.shape[1:], self._frame_shape))
        return rgb

    def preprocess_frames(self, rgb):
        """Environment._preprocess of n RGB frames up to the uint8 plane: [n,H,W,C] -> [n,84,84] uint8."""
        rgb = self._frames_arg(rgb)
        planes = np.empty((rgb.shape[0], Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH), np.uint8)
        nat.check(self._lib.ga3c_net_frames_preprocess(self._h, nat.ptr(rgb, nat.u8p), rgb.shape[0], nat.ptr(planes, nat.u8p)),
                  "ga3c_net_frames_preprocess")
        return planes

    def push_frames(self, rgb, agents, reset=None):
        """_update_frame_q for n distinct agents; reset[i] clears agent i's queue first (Environment.reset)."""
        rgb = self._frames_arg(rgb)
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        rs = None if reset is None else np.ascontiguousarray(reset, dtype=np.uint8)
        if agents.size != rgb.shape[0] or (rs is not None and rs.size != agents.size):
            raise ValueError("one agent id (and reset flag) per frame")
        seq = np.empty(agents.size, np.int64)
        nat.check(self._lib.ga3c_net_frames_push(self._h, nat.ptr(rgb, nat.u8p), nat.ptr(agents, nat.i32p),
                                                 None if rs is None else nat.ptr(rs, nat.u8p), agents.size,
                                                 nat.ptr(seq, nat.i64p)), "ga3c_net_frames_push")
        return seq

    def push_frame_offsets(self, offsets, agents, reset=None):
        """push_frames for frames lying in the registered transport (byte offsets), e.g. the agents' own slots."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        rs = None if reset is None else np.ascontiguousarray(reset, dtype=np.uint8)
        seq = np.empty(agents.size, np.int64)
        nat.check(self._lib.ga3c_net_frames_push_offsets(self._h, nat.ptr(offsets, nat.i64p), nat.ptr(agents, nat.i32p),
                                                         None if rs is None else nat.ptr(rs, nat.u8p), agents.size,
                                                         nat.ptr(seq, nat.i64p)), "ga3c_net_frames_push_offsets")
        return seq

    def train_frames(self, agents, seqs, y_r, a):
        """One training step on rows named by (agent, plane sequence number) -- see ga3c_net_train_frames."""
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        seqs = np.ascontiguousarray(seqs, dtype=np.int64)
        y, a = nat.as_f32(y_r), nat.as_f32(a)
        losses = np.empty(3, dtype=np.float32)
        # with a state cache the names are (agent, request number) of states the predictions stored; otherwise planes
        fn, name = ((self._lib.ga3c_net_train_cached, "ga3c_net_train_cached") if getattr(self, "_state_cache", False)
                    else (self._lib.ga3c_net_train_frames, "ga3c_net_train_frames"))
        nat.check(fn(self._h, nat.ptr(agents, nat.i32p), nat.ptr(seqs, nat.i64p), nat.ptr(y), nat.ptr(a), agents.size,
                     float(self.learning_rate), float(self.beta), nat.ptr(losses)), name)
        self.last_losses = losses

    def frame_state(self, agent):
        """(uint8 [84,84,4] state or None while the queue holds fewer than 4 planes, queue depth) -- _get_current_state."""
        state = np.empty((Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH, Config.STACKED_FRAMES), np.uint8)
        depth = C.c_int32()
        nat.check(self._lib.ga3c_net_frames_state(self._h, int(agent), nat.ptr(state, nat.u8p), C.byref(depth)),
                  "ga3c_net_frames_state")
        return (state if depth.value >= Config.STACKED_FRAMES else None), depth.value

    def frames_pushed(self, agent):
        """Planes pushed into `agent`'s device queue so far (= the sequence number its next plane gets)."""
        n = C.c_int64()
        nat.check(self._lib.ga3c_net_frames_pushed(self._h, int(agent), C.byref(n)), "ga3c_net_frames_pushed")
        return n.value

    def predict_frames(self, agents):
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        p = np.empty((agents.size, self.num_actions), dtype=np.float32)
        v = np.empty((agents.size,), dtype=np.float32)
        nat.check(self._lib.ga3c_net_predict_frames(self._h, nat.ptr(agents, nat.i32p), agents.size, nat.ptr(p), nat.ptr(v), None),
                  "ga3c_net_predict_frames")
        return [p, v]

    def frames_entry(self):
        """(address of ga3c_net_serve_frames, engine handle): the callback of the native raw-frame predictor loop."""
        return C.cast(self._lib.ga3c_net_serve_frames, C.c_void_p).value, self._h

    def frames_entries_pipelined(self):
        """(addresses of ga3c_net_serve_frames_begin / _end, engine handle) for ga3c_pq_serve_frames_pipelined."""
        return (C.cast(self._lib.ga3c_net_serve_frames_begin, C.c_void_p).value,
                C.cast(self._lib.ga3c_net_serve_frames_end, C.c_void_p).value, self._h)

    def serve_frames(self, offsets, agents, flags):
        """push + predict of one popped batch in one GPU round trip (what the native loop calls); rows of requests that
        asked for no prediction come back as zeros."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        flags = np.ascontiguousarray(flags, dtype=np.uint32)
        p = np.zeros((agents.size, self.num_actions), dtype=np.float32)
        v = np.zeros((agents.size,), dtype=np.float32)
        nat.check(self._lib.ga3c_net_serve_frames(self._h, nat.ptr(offsets, nat.i64p), nat.ptr(agents, nat.i32p),
                                                  nat.ptr(flags, nat.u32p), agents.size, nat.ptr(p), nat.ptr(v)),
                  "ga3c_net_serve_frames")
        return [p, v]

    def serve_frames_begin(self, offsets, agents, flags):
        """First half of serve_frames (ga3c_net_serve_frames_begin): frames pushed, forward pass enqueued -> ticket."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        agents = np.ascontiguousarray(agents, dtype=np.int32)
        flags = np.ascontiguousarray(flags, dtype=np.uint32)
        ticket = C.c_int32(-1)
        nat.check(self._lib.ga3c_net_serve_frames_begin(self._h, nat.ptr(offsets, nat.i64p), nat.ptr(agents, nat.i32p),
                                                        nat.ptr(flags, nat.u32p), agents.size, C.byref(ticket)),
                  "ga3c_net_serve_frames_begin")
        return ticket.value

    def serve_frames_end(self, ticket, flags):
        """Second half (ga3c_net_serve_frames_end): waits for the batch begun under `ticket` -> [p, v] as serve_frames."""
        flags = np.ascontiguousarray(flags, dtype=np.uint32)
        p = np.zeros((flags.size, self.num_actions), dtype=np.float32)
        v = np.zeros((flags.size,), dtype=np.float32)
        nat.check(self._lib.ga3c_net_serve_frames_end(self._h, int(ticket), nat.ptr(flags, nat.u32p), flags.size, nat.ptr(p),
                                                      nat.ptr(v)), "ga3c_net_serve_frames_end")
        return [p, v]

    def gather_entry(self):
        """(address of ga3c_net_predict_gather, engine handle, u8 flag): what the native predictor loop
        (ga3c_pq_serve, include/ga3c_host.h) calls for every batch instead of predict_offsets()."""
        return C.cast(self._lib.ga3c_net_predict_gather, C.c_void_p).value, self._h, int(self._transport_u8)

    def gather_entries_pipelined(self):
        """(addresses of ga3c_net_predict_gather_begin / _end, engine handle, u8 flag) for ga3c_pq_serve_pipelined."""
        return (C.cast(self._lib.ga3c_net_predict_gather_begin, C.c_void_p).value,
                C.cast(self._lib.ga3c_net_predict_gather_end, C.c_void_p).value, self._h, int(self._transport_u8))

    def state_cache_config(self, max_agents, depth):
        """Keep the uint8 states the pipelined predictor loop reads, `depth` per agent (ga3c_net_state_cache_config): rows of
        a train batch may then be named (agent, request number) -- train_frames / evaluate(frames=...) take such names."""
        nat.check(self._lib.ga3c_net_state_cache_config(self._h, int(max_agents), int(depth)), "ga3c_net_state_cache_config")
        self._state_cache = True

    def gather_entries_pipelined_cached(self):
        """As gather_entries_pipelined, with ga3c_net_predict_gather_begin_cached (None without a state cache)."""
        if not getattr(self, "_state_cache", False):
            return None
        return (C.cast(self._lib.ga3c_net_predict_gather_begin_cached, C.c_void_p).value,
                C.cast(self._lib.ga3c_net_predict_gather_end, C.c_void_p).value, self._h, int(self._transport_u8))

    def predict_offsets(self, offsets):
        """offsets: int64[B] byte offsets of the states inside the registered segment."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        b = offsets.size
        p = np.empty((b, self.num_actions), dtype=np.float32)
        v = np.empty((b,), dtype=np.float32)
        nat.check(self._lib.ga3c_net_predict_gather(self._h, nat.ptr(offsets, nat.i64p), b, int(self._transport_u8),
                                                    nat.ptr(p), nat.ptr(v), None), "ga3c_net_predict_gather")
        return [p, v]

    def train_offsets(self, offsets, y_r, a):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        y, a = nat.as_f32(y_r), nat.as_f32(a)
        losses = np.empty(3, dtype=np.float32)
        nat.check(self._lib.ga3c_net_train_gather(self._h, nat.ptr(offsets, nat.i64p), int(self._transport_u8), nat.ptr(y),
                                                  nat.ptr(a), offsets.size, float(self.learning_rate), float(self.beta),
                                                  nat.ptr(losses)), "ga3c_net_train_gather")
        self.last_losses = losses

    # ---- training ----------------------------------------------------------------------------
    def train(self, x, y_r, a, x2=None, done=None, trainer_id=0):
        """x2, done and trainer_id are accepted and ignored, as in NetworkVP.py:254-257."""
        y = nat.as_f32(y_r)             # arrives as float64 (ProcessAgent.py:99); cast at the boundary
        a = nat.as_f32(a)
        losses = np.empty(3, dtype=np.float32)
        if x.dtype == np.uint8:
            x = np.ascontiguousarray(x)
            nat.check(self._lib.ga3c_net_train_u8(self._h, nat.ptr(x, nat.u8p), nat.ptr(y), nat.ptr(a),
                                                  int(x.shape[0]), float(self.learning_rate), float(self.beta),
                                                  nat.ptr(losses)), "ga3c_net_train_u8")
        else:
            x = nat.as_f32(x)
            nat.check(self._lib.ga3c_net_train(self._h, nat.ptr(x), nat.ptr(y), nat.ptr(a), int(x.shape[0]),
                                               float(self.learning_rate), float(self.beta), nat.ptr(losses)),
                      "ga3c_net_train")
        self.last_losses = losses

    def compute_grads(self, x, y_r, a):
        x, y, a = nat.as_f32(x), nat.as_f32(y_r), nat.as_f32(a)
        losses = np.empty(3, dtype=np.float32)
        nat.check(self._lib.ga3c_net_compute_grads(self._h, nat.ptr(x), nat.ptr(y), nat.ptr(a), int(x.shape[0]),
                                                   float(self.beta), nat.ptr(losses)), "ga3c_net_compute_grads")
        return losses

    def apply_grads(self):
        nat.check(self._lib.ga3c_net_apply_grads(self._h, float(self.learning_rate)), "ga3c_net_apply_grads")

    def fetch(self, name, count):
        out = np.empty(int(count), dtype=np.float32)
        nat.check(self._lib.ga3c_net_fetch(self._h, name.encode(), nat.ptr(out), out.size), "ga3c_net_fetch")
        return out

    # ---- data-parallel ------------------------------------------------------------------------
    @staticmethod
    def make_comm_id():
        buf = np.zeros(nat.COMM_ID_BYTES, dtype=np.uint8)
        nat.check(nat.hip_lib().ga3c_comm_make_id(nat.ptr(buf, nat.u8p)), "ga3c_comm_make_id")
        return buf

    def comm_init(self, comm_id, rank, world):
        comm_id = np.ascontiguousarray(comm_id, dtype=np.uint8)
        nat.check(self._lib.ga3c_net_comm_init(self._h, nat.ptr(comm_id, nat.u8p), rank, world), "ga3c_net_comm_init")

    STAT_NAMES = ("predict_calls", "predict_rows", "predict_lane_wait_ns", "predict_launch_ns", "predict_sync_ns",
                  "predict_weight_waits", "train_calls", "train_rows", "train_stage_ns", "train_lane_wait_ns",
                  "train_launch_ns", "train_sync_ns", "train_reader_waits", "predict_gpu_ns", "state_cache_bytes",
                  "state_cache_lost_rows")

    def stats(self, reset=False):
        """Where the engine's calls spend their time (include/ga3c_abi.h: GA3C_STAT_*), as a dict of running totals."""
        out = np.zeros(len(self.STAT_NAMES), np.int64)
        nat.check(self._lib.ga3c_net_stats(self._h, nat.ptr(out, nat.i64p), out.size, 1 if reset else 0), "ga3c_net_stats")
        return dict(zip(self.STAT_NAMES, (int(v) for v in out)))

    def comm_info(self):
        """(ranks, rank, device) as the attached RCCL communicator reports them; (0, -1, -1) without one."""
        n, r, d = nat.C.c_int32(), nat.C.c_int32(), nat.C.c_int32()
        nat.check(self._lib.ga3c_net_comm_info(self._h, nat.C.byref(n), nat.C.byref(r), nat.C.byref(d)), "ga3c_net_comm_info")
        return n.value, r.value, d.value

    # ---- logging / checkpoints -----------------------------------------------------------------
    def evaluate(self, x, y_r, a, offsets=None, frames=None):
        """Forward + loss of the batch on the current weights, no update: what sess.run(summary_op) evaluates
        (NetworkVP.py:259-265).  The states are `x` (f32 or uint8 [B,84,84,4]), or rows still lying in the registered
        transport (`offsets`), or rows named by (agents, plane sequence numbers) (`frames`).
        Returns (losses[3], d1[B,256], v[B], p[B,A])."""
        y, a = nat.as_f32(y_r), nat.as_f32(a)
        b = int(y.shape[0])
        losses = np.empty(3, np.float32)
        d1 = np.empty((b, 256), np.float32)
        v = np.empty(b, np.float32)
        p = np.empty((b, self.num_actions), np.float32)
        outs = (nat.ptr(losses), nat.ptr(d1), nat.ptr(v), nat.ptr(p))
        if frames is not None:
            agents = np.ascontiguousarray(frames[0], dtype=np.int32)
            seqs = np.ascontiguousarray(frames[1], dtype=np.int64)
            fn, name = ((self._lib.ga3c_net_evaluate_cached, "ga3c_net_evaluate_cached") if getattr(self, "_state_cache", False)
                        else (self._lib.ga3c_net_evaluate_frames, "ga3c_net_evaluate_frames"))
            nat.check(fn(self._h, nat.ptr(agents, nat.i32p), nat.ptr(seqs, nat.i64p), nat.ptr(y), nat.ptr(a), b,
                         float(self.beta), *outs), name)
        elif offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.int64)
            nat.check(self._lib.ga3c_net_evaluate(self._h, None, None, nat.ptr(offsets, nat.i64p), int(self._transport_u8),
                                                  nat.ptr(y), nat.ptr(a), b, float(self.beta), *outs), "ga3c_net_evaluate")
        elif x.dtype == np.uint8:
            x = np.ascontiguousarray(x)
            nat.check(self._lib.ga3c_net_evaluate(self._h, None, nat.ptr(x, nat.u8p), None, 0, nat.ptr(y), nat.ptr(a), b,
                                                  float(self.beta), *outs), "ga3c_net_evaluate")
        else:
            x = nat.as_f32(x)
            nat.check(self._lib.ga3c_net_evaluate(self._h, nat.ptr(x), None, None, 0, nat.ptr(y), nat.ptr(a), b,
                                                  float(self.beta), *outs), "ga3c_net_evaluate")
        return losses, d1, v, p

    def log(self, x, y_r, a, training_step, feed_dict=None, offsets=None, frames=None):
        """The reference's summary_op on the batch it is given (NetworkVP.py:259-265, NetworkVP_discrate.py:132-151):
        the six scalars (Pcost_advantage, Pcost_entropy, Pcost, Vcost, LearningRate, Beta) appended to
        logs/<model>/scalars.csv, and the histograms (one per trainable variable, activation_lastdense, activation_v,
        activation_p) written to logs/<model>/histograms_%08d.npz with the fields of TensorFlow's HistogramProto."""
        losses, d1, v, p = self.evaluate(x, y_r, a, offsets=offsets, frames=frames)
        c1, c2, cv = (float(t) for t in losses)
        theta = self.get_arena(0)
        hist = {}
        for name in PARAM_ORDER:
            off, size = self._offsets[name]
            hist["weights_%s:0" % name] = histogram_proto(theta[off:off + size])
        hist["activation_lastdense"] = histogram_proto(d1)
        hist["activation_v"] = histogram_proto(v)
        hist["activation_p"] = histogram_proto(p)
        out = {}
        for tag, h in hist.items():
            for field, value in h.items():
                out["%s/%s" % (tag, field)] = value
        os.makedirs("logs/%s" % self.model_name, exist_ok=True)
        with self._log_lock:
            with open("logs/%s/scalars.csv" % self.model_name, "a") as f:
                f.write("%d,%.8g,%.8g,%.8g,%.8g,%.8g,%.8g\n" % (training_step, c1, c2, -(c1 + c2), cv,
                                                                self.learning_rate, self.beta))
            tmp = "logs/%s/histograms_%08d.tmp.npz" % (self.model_name, training_step)
            np.savez(tmp, **out)
            os.replace(tmp, "logs/%s/histograms_%08d.npz" % (self.model_name, training_step))
        return losses

    def _checkpoint_filename(self, episode):
        return 'checkpoints/%s_%08d' % (self.model_name, episode)

    def _get_episode_from_filename(self, filename):
        return int(re.split(r'/|_|\.', filename)[2])

    def save(self, episode):
        """Own on-disk format (.npz keyed by the TF variable names + RMSProp slots + step):
        a TF checkpoint cannot be written without TF."""
        os.makedirs("checkpoints", exist_ok=True)
        # written by the library itself (ga3c_net_save: an uncompressed .npz, written under a temporary name and renamed)
        nat.check(self._lib.ga3c_net_save(self._h, (self._checkpoint_filename(episode) + ".npz").encode()), "ga3c_net_save")

    def load(self):
        if Config.LOAD_EPISODE > 0:
            filename = self._checkpoint_filename(Config.LOAD_EPISODE) + ".npz"
        else:
            found = sorted(glob.glob('checkpoints/%s_????????.npz' % self.model_name))
            if not found:
                raise FileNotFoundError("no checkpoint for %s" % self.model_name)
            filename = found[-1]
        nat.check(self._lib.ga3c_net_load(self._h, filename.encode()), "ga3c_net_load")
        return self._get_episode_from_filename(filename[:-4])
