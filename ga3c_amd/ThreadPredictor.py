"""Dynamic batcher for inference (reference: ga3c/ThreadPredictor.py:34-66).

Same policy: block for one request, then drain whatever else is queued up to
PREDICTION_BATCH_SIZE WITHOUT waiting, run one forward, hand (p[i], v[i]) back to agent ids[i].
Differences forced by the transport: the blocking pop times out every QUEUE_TIMEOUT_MS so that
remove_predictor() cannot hang on an idle queue (the reference's :50 can); ids are u32, and the
network-tester id 100 special case (:64-66) is not reproduced.

With the zero-copy transport and a model that offers gather_entry() the loop itself runs in native code
(ga3c_pq_serve, include/ga3c_host.h: the same pop -> offsets -> ga3c_net_predict_gather -> respond steps), in
time slices of SERVE_SLICE_MS so that this thread holds the interpreter lock only to look at exit_flag and fold the
counters; Config.NATIVE_PREDICTOR = False keeps the Python loop below.
"""
import os
import time
from threading import Thread

import numpy as np

from Config import Config
import _native as nat

SERVE_SLICE_MS = 50


class ThreadPredictor(Thread):
    def __init__(self, server, id, state_dim, transport):
        super(ThreadPredictor, self).__init__()
        self.daemon = True
        self.id = id
        self.server = server
        self.state_dim = state_dim
        self.transport = transport
        self.exit_flag = False
        self.batches = 0
        self.served = 0
        self.seconds = {"pop": 0.0, "predict": 0.0, "respond": 0.0}     # where the loop's wall time went
        self.native = False

    def _run_native(self, entry, handle, u8):
        t, st = self.transport, nat.ServeStats()
        self.native = True
        # answering batch k beside the GPU's work on batch k+1 (ga3c_pq_serve_pipelined) when the model offers the split call
        split = getattr(self.server.model, "gather_entries_pipelined", None) if getattr(Config, "PIPELINED_PREDICTOR", True) else None
        begin_end = split() if split else None
        # ... and, with the engine's state cache in use, telling it each row's name (agent, request number)
        cached = getattr(self.server.model, "gather_entries_pipelined_cached", None) if getattr(self.server, "state_cache", False) else None
        cached = cached() if cached else None
        while not self.exit_flag:
            if cached:
                rc = t.serve_pipelined_cached(cached[0], cached[1], handle, u8, Config.PREDICTION_BATCH_SIZE, SERVE_SLICE_MS, st)
            elif begin_end:
                rc = t.serve_pipelined(begin_end[0], begin_end[1], handle, u8, Config.PREDICTION_BATCH_SIZE, SERVE_SLICE_MS, st)
            else:
                rc = t.serve(entry, handle, u8, Config.PREDICTION_BATCH_SIZE, SERVE_SLICE_MS, st)
            self.batches, self.served = st.batches, st.served
            self.seconds = {"pop": st.ns_pop * 1e-9, "predict": st.ns_predict * 1e-9, "respond": st.ns_respond * 1e-9}
            if rc < 0:
                break                                   # transport shut down

    def _run_frames(self):
        """Device front-end: the popped slots hold raw emulator frames.  Push them into the agents' frame queues (the
        GPU reads them in place), then answer with predictions for the agents whose queue the frame completed."""
        import Transport as tp
        t, model = self.transport, self.server.model
        entry = getattr(model, "frames_entry", None)
        if entry and getattr(Config, "NATIVE_PREDICTOR", True):     # the same loop in native code, one GPU round trip per batch
            fn, handle = entry()
            st = nat.ServeStats()
            self.native = True
            # Config.PIPELINED_FRAMES: answering batch k beside the GPU's work on batch k+1, the model's call in two halves
            # (ga3c_pq_serve_frames_pipelined; off by default -- it pays from ~500 agents per GPU on, DESIGN.md section 5;
            # the helper-thread modes of GA3C_RESPONDER belong to the one-piece loop)
            on = os.environ.get("GA3C_PIPELINE_FRAMES")
            on = bool(int(on)) if on not in (None, "") else bool(getattr(Config, "PIPELINED_FRAMES", False))
            split = getattr(model, "frames_entries_pipelined", None) if on else None
            if split and os.environ.get("GA3C_RESPONDER", "0") not in ("0", "3", ""):
                split = None
            halves = split() if split else None
            while not self.exit_flag:
                if halves:
                    rc = t.serve_frames_pipelined(halves[0], halves[1], halves[2], Config.PREDICTION_BATCH_SIZE, SERVE_SLICE_MS, st)
                else:
                    rc = t.serve_frames(fn, handle, Config.PREDICTION_BATCH_SIZE, SERVE_SLICE_MS, st)
                self.batches, self.served = st.batches, st.served
                self.seconds = {"pop": st.ns_pop * 1e-9, "predict": st.ns_predict * 1e-9, "respond": st.ns_respond * 1e-9}
                if rc < 0:
                    break
            return
        bmax = Config.PREDICTION_BATCH_SIZE
        ids = np.zeros(bmax, dtype=np.uint32)
        p = np.zeros((bmax, t.num_actions), np.float32)
        v = np.zeros(bmax, np.float32)
        clock, spent = time.perf_counter, self.seconds
        while not self.exit_flag:
            t0 = clock()
            size = t.pop_batch(ids, Config.QUEUE_TIMEOUT_MS)
            t1 = clock()
            spent["pop"] += t1 - t0
            if size == 0:
                continue
            if size < 0:
                break
            got = ids[:size]
            flags = t.request_flags(got)
            model.push_frame_offsets(t.state_offsets(got), got, (flags & tp.REQ_RESET) != 0)
            want = np.nonzero((flags & tp.REQ_NO_PREDICT) == 0)[0]
            if want.size:
                pw, vw = model.predict_frames(got[want])
                p[want], v[want] = pw, vw
            t2 = clock()
            t.respond(ids, size, p, v)
            spent["predict"] += t2 - t1
            spent["respond"] += clock() - t2
            self.batches += 1
            self.served += int(want.size)

    def run(self):
        """The loop of the reference's run(); a failure is reported to the server instead of dying with the thread."""
        try:
            self._run()
        except BaseException as e:   # noqa: BLE001
            report = getattr(self.server, "worker_failed", None)
            if report is None:
                raise
            report("%s %d" % (type(self).__name__, self.id), e)

    def _run(self):
        t = self.transport
        if getattr(self.server, "device_frontend", False):
            return self._run_frames()
        entry = getattr(self.server.model, "gather_entry", None)
        if entry and getattr(self.server, "zero_copy", False) and getattr(Config, "NATIVE_PREDICTOR", True):
            return self._run_native(*entry())
        bmax = Config.PREDICTION_BATCH_SIZE
        ids = np.zeros(bmax, dtype=np.uint32)
        u8 = t.state_bytes == int(np.prod(self.state_dim))
        # staging batch in pinned memory when the model offers it, so the H2D copy is a plain DMA
        alloc = getattr(self.server.model, "pinned_array", None)
        shape = (bmax, t.state_bytes)
        staging = None
        if not getattr(self.server, "zero_copy", False):
            staging = alloc(shape, np.uint8) if alloc else np.zeros(shape, np.uint8)
        zero_copy = getattr(self.server, "zero_copy", False)
        clock, spent = time.perf_counter, self.seconds
        while not self.exit_flag:
            t0 = clock()
            size = t.pop_batch(ids, Config.QUEUE_TIMEOUT_MS)
            t1 = clock()
            spent["pop"] += t1 - t0
            if size == 0:
                continue
            if size < 0:
                break                                   # transport shut down
            if zero_copy:                               # the GPU gathers the states out of the slots itself
                p, v = self.server.model.predict_offsets(t.state_offsets(ids[:size]))
            else:
                np.take(t.agent_states, ids[:size], axis=0, out=staging[:size])
                batch = staging[:size] if u8 else staging[:size].view(np.float32)
                p, v = self.server.model.predict_p_and_v(batch.reshape((size,) + tuple(self.state_dim)))
            t2 = clock()
            t.respond(ids, size, np.ascontiguousarray(p, np.float32), np.ascontiguousarray(v, np.float32))
            spent["predict"] += t2 - t1
            spent["respond"] += clock() - t2
            self.batches += 1
            self.served += size
