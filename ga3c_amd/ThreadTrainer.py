"""Training batcher (reference: ga3c/ThreadTrainer.py:33-62): take rollouts from the training
queue until the batch holds MORE than TRAINING_MIN_BATCH_SIZE rows, then one server.train_model().

Rollouts arrive as slots of the shared-memory transport (states, f32 returns, int32 actions);
rows are copied once into a staging batch instead of the reference's repeated np.concatenate.
x2_ and done_ are not transported (unused by the A3C nets, NetworkVP.py:254); train_model gets None.

Zero-copy intake keeps a rollout's slot until the GPU has read its rows, which the reference's queue never does
(training_q.get() frees the entry).  Two rules keep that from starving the agents of slots: only one trainer at a time
fills a batch (Server.batch_lock; the others are training or waiting their turn), and a trainer that holds slots while
the agents have none left and nothing is queued SPILLS: it copies the rows it holds into a host batch, gives the slots
back and finishes that batch through the host-buffer path.  (Seen as a dead stop with TRAINING_MIN_BATCH_SIZE = 511,
MAX_QUEUE_SIZE = 100 and two trainers.)
"""
from threading import Thread

import numpy as np

from Config import Config


class ThreadTrainer(Thread):
    def __init__(self, server, id, transport=None):
        super(ThreadTrainer, self).__init__()
        self.daemon = True
        self.id = id
        self.server = server
        self.transport = transport if transport is not None else server.transport
        self.exit_flag = False
        self.spills = 0                 # batches finished through the host path because the agents ran out of slots

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
        cap = Config.TRAINING_MIN_BATCH_SIZE + t.train_rows
        state_dim = tuple(self.server.state_dim)
        u8 = t.state_bytes == int(np.prod(state_dim))
        alloc = getattr(self.server.model, "pinned_array", None)
        shape = (cap, t.state_bytes)
        zero_copy = getattr(self.server, "zero_copy", False)
        # rows name states kept in HBM: (plane number, agent) with the frame queue on the device, (request number, agent) with
        # the engine's state cache
        on_device = getattr(self.server, "device_frontend", False) or getattr(self.server, "state_cache", False)
        seq_stage = np.zeros(cap, np.int64)
        agent_stage = np.zeros(cap, np.int32)
        x_stage = None
        if not zero_copy and not on_device:
            x_stage = alloc(shape, np.uint8) if alloc else np.zeros(shape, np.uint8)
        off_stage = np.zeros(cap, np.int64)
        r_stage = np.zeros(cap, np.float32)
        a_stage = np.zeros(cap, np.int32)
        eye = np.eye(t.num_actions, dtype=np.float32)
        holding = zero_copy and not on_device
        turn = getattr(self.server, "batch_lock", None) if holding else None
        # zero-copy batches are assembled by ONE native call (ga3c_tq_collect: the loop below, minus the interpreter --
        # at ~26 rollouts per 128-row batch the Python loop was 350 us under batch_lock, the trainers' ceiling)
        native = (holding or on_device) and getattr(Config, "NATIVE_TRAINER", True) and hasattr(t, "collect")
        cstate = np.zeros(2, np.int32)                  # rows, slots of the batch in progress
        slot_stage = np.zeros(cap + 1, np.int32)
        while not self.exit_flag:
            batch_size = 0
            held = []                                   # zero-copy: (slot, first row, rows) stay ours until the GPU has read them
            spilled = False
            starved = False
            cstate[:] = 0
            if turn:
                while not turn.acquire(timeout=Config.QUEUE_TIMEOUT_MS / 1000.0):
                    if self.exit_flag:
                        return
            try:
                while batch_size <= Config.TRAINING_MIN_BATCH_SIZE and not self.exit_flag:
                    if native and not spilled:
                        rc = t.collect(Config.TRAINING_MIN_BATCH_SIZE, Config.QUEUE_TIMEOUT_MS, 5, cstate, slot_stage, off_stage,
                                       r_stage, a_stage, seq_stage if on_device else None, agent_stage if on_device else None)
                        batch_size = int(cstate[0])
                        if rc == -4:
                            return                      # transport shut down
                        if rc == 1:
                            # starved: rebuild the held list and spill.  The native call has already seen "nothing free,
                            # nothing queued" while these slots were held; the counts are NOT looked at again here -- an
                            # agent that commits in between would skip the spill and leave the rows of `held` to be
                            # overwritten by the next collect (round-2 advice)
                            starved = True
                            first = 0
                            for slot in slot_stage[:cstate[1]]:
                                rows = t.rows(int(slot))
                                held.append((int(slot), first, rows))
                                first += rows
                            cstate[1] = 0
                        else:
                            continue                    # complete (the loop condition ends it) or timeout
                    if held and (starved or (t.free_count() == 0 and t.ready_count() == 0)):
                        starved = False
                        # every slot is ours or another trainer's and the agents are waiting for one: spill
                        if x_stage is None:
                            x_stage = alloc(shape, np.uint8) if alloc else np.zeros(shape, np.uint8)
                        for slot, first, rows in held:
                            x_stage[first:first + rows] = t.rollout_views(slot)[0][:rows]
                            t.release(slot)
                        held, spilled = [], True
                        self.spills += 1
                    slot = t.pop_rollout(5 if held else Config.QUEUE_TIMEOUT_MS)
                    if slot == -3:
                        continue                        # timeout: look at exit_flag (and at the slot supply) again
                    if slot < 0:
                        return                          # transport shut down
                    rows = t.rows(slot)
                    states, returns, actions = t.rollout_views(slot)
                    r_stage[batch_size:batch_size + rows] = returns[:rows]
                    a_stage[batch_size:batch_size + rows] = actions[:rows]
                    if on_device:
                        seq_stage[batch_size:batch_size + rows] = states[:rows, :8].view(np.int64).ravel()
                        agent_stage[batch_size:batch_size + rows] = states[:rows, 8:12].view(np.int32).ravel()
                        t.release(slot)
                    elif holding and not spilled:
                        off_stage[batch_size:batch_size + rows] = t.rollout_row_offsets(slot, rows)
                        held.append((slot, batch_size, rows))
                    else:
                        x_stage[batch_size:batch_size + rows] = states[:rows]
                        t.release(slot)
                    batch_size += rows
            finally:
                if turn:
                    turn.release()
            if batch_size and Config.TRAIN_MODELS and not self.exit_flag:
                if on_device:
                    self.server.train_model_frames(agent_stage[:batch_size], seq_stage[:batch_size], r_stage[:batch_size],
                                                   eye[a_stage[:batch_size]], self.id)
                elif holding and not spilled:
                    self.server.train_model_rows(off_stage[:batch_size], r_stage[:batch_size],
                                                 eye[a_stage[:batch_size]], self.id)
                else:
                    xb = x_stage[:batch_size] if u8 else x_stage[:batch_size].view(np.float32)
                    self.server.train_model(xb.reshape((batch_size,) + state_dim), r_stage[:batch_size],
                                            eye[a_stage[:batch_size]], None, None, self.id)
            for slot, _, _ in held:
                t.release(slot)
            if native and cstate[1]:
                t.release_many(slot_stage, cstate[1])
            if self.exit_flag or batch_size == 0:
                break
