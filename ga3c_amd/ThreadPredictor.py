"""Dynamic batcher for inference (reference: ga3c/ThreadPredictor.py:34-66).

Same policy: block for one request, then drain whatever else is queued up to
PREDICTION_BATCH_SIZE WITHOUT waiting, run one forward, hand (p[i], v[i]) back to agent ids[i].
Differences forced by the transport: the blocking pop times out every QUEUE_TIMEOUT_MS so that
remove_predictor() cannot hang on an idle queue (the reference's :50 can); ids are u32, and the
network-tester id 100 special case (:64-66) is not reproduced.
"""
from threading import Thread

import numpy as np

from Config import Config


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

    def run(self):
        t = self.transport
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
        while not self.exit_flag:
            size = t.pop_batch(ids, Config.QUEUE_TIMEOUT_MS)
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
            t.respond(ids, size, np.ascontiguousarray(p, np.float32), np.ascontiguousarray(v, np.float32))
            self.batches += 1
            self.served += size
            self.server.predictions_served += size
