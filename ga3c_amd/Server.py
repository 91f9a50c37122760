"""Orchestrator: owns the transport, the model and the worker lists; counts train steps; anneals
learning rate and beta; services checkpoint requests (reference: ga3c/Server.py:69-206).

What changed against the reference, and why:
  * prediction_q / training_q / wait_q are one shared-memory Transport (Transport.py) created
    before any agent starts;
  * the model is the HIP-backed Network (NetworkVP.py), selected where the reference selects its
    TensorFlow class (Server.py:48-54);
  * agents are started from a forkserver, so they never inherit the server's HIP state.
"""
import os
import threading
import time

from Config import Config
import DataParallel
from Environment import Environment
from NetworkVP import Network, _device_ordinal
import Placement
import _native as nat
from ProcessAgent import ProcessAgent, config_snapshot
from ProcessStats import ProcessStats
from ThreadDynamicAdjustment import ThreadDynamicAdjustment
from ThreadPredictor import ThreadPredictor
from ThreadTrainer import ThreadTrainer
import Transport as tp


class Server:
    def __init__(self, model=None, max_agents=None, engine_group=None):
        # one Server per GPU under torch.distributed.run: lock-step training over RCCL (DataParallel.py)
        self.dp = engine_group
        self.dp_lock = threading.Lock()
        self.batch_lock = threading.Lock()      # one trainer at a time fills a batch (ThreadTrainer.py)
        self.closing = False
        # before anything is started: every thread and process created from here on inherits the placement (Placement.py)
        self.placement = Placement.place(os.environ.get("GA3C_CPU_AFFINITY") or getattr(Config, "CPU_AFFINITY", "auto"),
                                         _device_ordinal(Config.DEVICE))
        Config.AGENT_CPUS = (self.placement or {}).get("agent_cpus")
        self.stats = ProcessStats()
        self.state_dim = self.get_state_dim()
        self.num_actions = self.get_num_action()
        self.max_agents = int(max_agents or max(2 * Config.AGENTS, Config.AGENTS + 16))
        n_state = Config.IMAGE_HEIGHT * Config.IMAGE_WIDTH * Config.STACKED_FRAMES
        state_bytes = n_state if Config.STATE_TRANSPORT == 'u8' else 4 * n_state
        # device-side frame front-end: slots carry the emulator's raw frame, rollout rows only name their state
        self.device_frontend = Config.FRONTEND == 'device'
        raw = Config.FRAME_SOURCE in ('rgb', 'gym')         # raw emulator frames; otherwise ready-made 84x84 planes
        self.frame_shape = (Config.FRAME_HEIGHT, Config.FRAME_WIDTH, 3) if raw else (Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH, 1)
        row_bytes = 0
        if self.device_frontend:
            state_bytes = (self.frame_shape[0] * self.frame_shape[1] * self.frame_shape[2] + 15) // 16 * 16
            row_bytes = 16
        # State cache: the engine keeps the uint8 states its predictions read and rollouts name them (agent, request number)
        # instead of carrying them -- a trainer's batch no longer crosses PCIe a second time (include/ga3c_abi.h).  Only with
        # everything it rests on: the GPU reading the transport itself, uint8 states, the native pipelined predictor loop,
        # plain launches; anything else keeps the states in the rollouts.
        model_cls = Network if model is None else type(model)
        self.state_cache = bool(getattr(Config, "STATE_CACHE", False) and not self.device_frontend and Config.ZERO_COPY and
                                Config.STATE_TRANSPORT == 'u8' and getattr(Config, "NATIVE_PREDICTOR", True) and
                                getattr(Config, "PIPELINED_PREDICTOR", True) and getattr(Config, "NATIVE_TRAINER", True) and
                                not os.environ.get("GA3C_GRAPHS") and
                                hasattr(model_cls, "state_cache_config") and hasattr(model_cls, "gather_entries_pipelined_cached"))
        Config.STATE_CACHE_ACTIVE = self.state_cache         # (the agents read it from their configuration snapshot)
        if self.state_cache:
            row_bytes = 16
        self.model = model if model is not None else Network(Config.DEVICE, Config.NETWORK_NAME, self.num_actions,
                                                             self.state_dim)
        # training_q.get() frees a queue entry at once (ThreadTrainer.py:49); zero-copy trainers keep a rollout's slot
        # until the GPU has read it, so the slots they hold come on top of the queue bound
        slots = int(Config.ROLLOUT_SLOTS)

        def slot_count():
            per_batch = -(-(Config.TRAINING_MIN_BATCH_SIZE + 1) // max(Config.TIME_MAX, 1))
            return Config.MAX_QUEUE_SIZE + (0 if (self.device_frontend or self.state_cache) else 2 * max(Config.TRAINERS, 2) * per_batch)
        if self.state_cache:
            # What the ring of an agent must hold: the states it has stored and not yet seen trained.  Over ALL agents that is
            # bounded by the rollouts in flight (every slot of the transport, two more being filled or handed over) plus the rows
            # the trainers hold after they have given the slots back -- with the trainer count the dynamic adjustment may reach,
            # not today's.  One agent owning all of it (886 states = 25 MB at the defaults, x max_agents) does not happen with
            # agents that step at one rate: the ring gets four times the agent's fair share of that bound plus four rollouts,
            # at least 64 states (Config.STATE_CACHE_DEPTH overrides); a row that does fall out is refused by the engine
            # (GA3C_ELOST) and its batch dropped and counted, never trained on wrong bytes.
            trainers = max(Config.TRAINERS, 2, 8 if Config.DYNAMIC_SETTINGS else 0)
            total = ((slots if slots > 0 else slot_count()) + 2) * (Config.TIME_MAX + 1) + 8 + \
                trainers * (Config.TRAINING_MIN_BATCH_SIZE + Config.TIME_MAX + 1)
            # (with the dynamic adjustment on the live agents can be a fraction of max_agents and each one's share of what is
            # in flight that much larger: eight times the fair share then -- a 330-s soak at four lost one batch of 1.4 M)
            share = 8 if Config.DYNAMIC_SETTINGS else 4
            depth = int(getattr(Config, "STATE_CACHE_DEPTH", 0)) or \
                min(total, max(64, share * -(-total // self.max_agents) + 4 * (Config.TIME_MAX + 1)))
            try:
                self.model.state_cache_config(self.max_agents, depth)
                self.state_cache_depth = depth
            except RuntimeError as e:            # no room in HBM: rollouts carry their states, as without the cache
                print("[state cache] off: %s" % e, flush=True)
                self.state_cache = False
                Config.STATE_CACHE_ACTIVE = False
                row_bytes = 16 if self.device_frontend else 0
        if slots <= 0:
            slots = slot_count()
        self.transport = tp.Transport.create(tp.unique_name(), self.max_agents, self.num_actions, state_bytes,
                                             slots, Config.TIME_MAX + 1, row_bytes)
        if getattr(Config, "AGENT_SPIN_US", 0) > 0:
            self.transport.set_spin(Config.AGENT_SPIN_US)
        if Config.PREDICTION_LINGER_US > 0:
            self.transport.set_linger(Config.PREDICTION_LINGER_US, Config.PREDICTION_LINGER_BATCH)
        if self.dp is not None and hasattr(self.model, "comm_init"):
            DataParallel.attach(self.model, self.dp.rank, self.dp.world, rendezvous=self.dp.rv)
        # let the GPU read states straight out of the transport's slots (no host gather, no staging copy)
        self.zero_copy = bool(Config.ZERO_COPY) and hasattr(self.model, "register_transport")
        if self.zero_copy:
            self.model.register_transport(self.transport)
        if self.device_frontend:
            if not (self.zero_copy and hasattr(self.model, "frames_config")):
                raise RuntimeError("FRONTEND = 'device' needs ZERO_COPY and a model with the frames_* entry points")
            # an agent can be ahead of the trainers by every rollout in flight plus the one it is filling
            # ... and a trainer gives a rollout's slot back when it has copied the row names, BEFORE the batch is trained:
            # every trainer can hold TRAINING_MIN_BATCH_SIZE + TIME_MAX + 1 such rows, in the worst case of one agent
            history = Config.FRAME_HISTORY or ((self.transport.train_slots + 2) * (Config.TIME_MAX + 1) + 8 +
                                               max(Config.TRAINERS, 2) * (Config.TRAINING_MIN_BATCH_SIZE + Config.TIME_MAX + 1))
            self.model.frames_config(self.max_agents, self.frame_shape[0], self.frame_shape[1], self.frame_shape[2], history)
        if self.state_cache and not self.zero_copy:
            raise RuntimeError("STATE_CACHE needs a model that reads the transport itself (register_transport)")
        if Config.LOAD_CHECKPOINT:
            try:
                self.stats.episode_count.value = self.model.load()
            except (OSError, KeyError, ValueError, RuntimeError) as e:
                print("checkpoint not loaded: %s" % e)
        self.training_step = 0
        self.lost_train_batches = 0              # batches dropped because a named state had left the state cache
        self.frame_counter = 0
        self._served_by_retired = 0
        self.agents = []
        self.agent_id = 0                        # next never-used id
        self.free_agent_ids = []                 # ids of removed agents, reusable once their last request is answered
        self.failure = None                      # (worker name, exception) of the first batching thread that died
        self.predictors = []
        self.trainers = []
        self.dynamic_adjustment = ThreadDynamicAdjustment(self)
        print("Server initialized")

    # ---- worker lifecycle (Server.py:106-139) ---------------------------------------------------
    def add_agent(self):
        """Server.py:106-110.  Agent ids index the transport's slots, so the ids of removed agents are reused (oldest
        first) once their last request has been answered; a new id is taken only while there are slots left.  With the
        random walk of ThreadDynamicAdjustment adding and removing agents for hours, ids would otherwise run out."""
        agent_id = None
        for k, cand in enumerate(self.free_agent_ids):
            if self.transport.agent_idle(cand):
                agent_id = self.free_agent_ids.pop(k)
                break
        if agent_id is None:
            if self.agent_id >= self.max_agents:
                return False
            agent_id = self.agent_id
            self.agent_id += 1
        planes = self.model.frames_pushed(agent_id) if (self.device_frontend and hasattr(self.model, "frames_pushed")) else 0
        self.agents.append(ProcessAgent(agent_id, self.transport.name, self.stats.episode_log_q, config_snapshot(), planes))
        self.agents[-1].start()
        return True

    def remove_agent(self):
        agent = self.agents[-1]
        agent.exit_flag.value = True
        agent.join(5)
        if agent.is_alive():
            agent.terminate()
            agent.join(5)
        self.agents.pop()
        self.free_agent_ids.append(agent.id)

    def add_predictor(self):
        self.predictors.append(ThreadPredictor(self, len(self.predictors), self.state_dim, self.transport))
        self.predictors[-1].start()

    def remove_predictor(self):
        self.predictors[-1].exit_flag = True
        self.predictors[-1].join()
        self._served_by_retired += self.predictors.pop().served

    @property
    def predictions_served(self):
        return self._served_by_retired + sum(p.served for p in self.predictors)

    def add_trainer(self):
        self.trainers.append(ThreadTrainer(self, len(self.trainers)))
        self.trainers[-1].start()

    def remove_trainer(self):
        self.trainers[-1].exit_flag = True
        # after a failure a trainer may sit inside a collective its peers never join: it is a daemon thread, leave it
        self.trainers[-1].join(None if self.failure is None else 5.0)
        self.trainers.pop()

    # ---- training bookkeeping (Server.py:141-153) ----------------------------------------------
    def _dp_gate(self):
        """Data-parallel runs (call with dp_lock held): wait until rank 0's credit covers the next step, then set the
        learning rate and beta that rank 0 attached to that step.  False = the group is stopping and this rank has
        taken its last step (the batch is dropped, as the reference drops what is queued at exit)."""
        dp = self.dp
        t0 = time.time()
        while not dp.may_step(self.training_step):
            if dp.finished(self.training_step) or self.closing or self.failure is not None:
                return False
            if time.time() - t0 > dp.STALL_S:
                # bounded: the main loop's poll() normally reports this first (and names the rank); a trainer thread that
                # has waited this long for credit reports it itself rather than stall for ever
                self.worker_failed("data-parallel group", DataParallel.GroupStalled(
                    "rank %d: no credit for train step %d from rank 0 for %.0f s" % (dp.rank, self.training_step + 1, time.time() - t0)))
                return False
            time.sleep(0.0005)
        self.model.learning_rate, self.model.beta = dp.rates_for(self.training_step + 1)
        dp.note_started(self.training_step + 1)
        return True

    def train_model(self, x_, r_, a_, x2, done, trainer_id):
        if self.dp is None:             # trainer threads go straight to the model, as in Server.py:141-142
            self.model.train(x_, r_, a_, x2, done, trainer_id)
            self._count_train_step(x_.shape[0], x_, r_, a_)
            return
        with self.dp_lock:
            if self._dp_gate():
                self.model.train(x_, r_, a_, x2, done, trainer_id)
                self._count_train_step(x_.shape[0], x_, r_, a_)

    def _count_train_step(self, rows, x_, r_, a_, **where):
        self.training_step += 1
        self.frame_counter += rows
        self.stats.training_count.value += 1
        self.dynamic_adjustment.temporal_training_count += 1
        if Config.TENSORBOARD and self.stats.training_count.value % Config.TENSORBOARD_UPDATE_FREQUENCY == 0:
            try:
                self.model.log(x_, r_, a_, self.training_step, **where)     # the batch just trained (Server.py:149-150)
            except nat.StateLost:               # a named row left the state cache since the step: no summary this time
                pass

    def worker_failed(self, worker, exc):
        """A predictor / trainer thread died (a HIP error, a row that left the plane history, ...).  The reference lets
        such a thread die silently and keeps running without it (Errors); here the first failure
        is kept, main() stops the server and re-raises it, so the process ends with a non-zero status."""
        if self.failure is None:
            self.failure = (worker, exc)

    def train_model_rows(self, row_offsets, r_, a_, trainer_id):
        """train_model for rows that are still sitting in the transport (zero-copy intake)."""
        if self.dp is None:
            self.model.train_offsets(row_offsets, r_, a_)
            self._count_train_step(row_offsets.shape[0], None, r_, a_, offsets=row_offsets)
            return
        with self.dp_lock:
            if self._dp_gate():
                self.model.train_offsets(row_offsets, r_, a_)
                self._count_train_step(row_offsets.shape[0], None, r_, a_, offsets=row_offsets)

    def train_model_frames(self, agents, seqs, r_, a_, trainer_id):
        """train_model for rows whose states live in the device-side plane history (FRONTEND = 'device')."""
        if self.dp is None:
            try:
                self.model.train_frames(agents, seqs, r_, a_)
            except nat.StateLost as e:          # a row fell out of the state cache (GA3C_ELOST): nothing was trained, the batch
                self.lost_train_batches += 1    # is dropped and counted; the first one says how to make room
                if self.lost_train_batches == 1:
                    print("[state cache] %s" % e, flush=True)
                return
            self._count_train_step(agents.shape[0], None, r_, a_, frames=(agents, seqs))
            return
        with self.dp_lock:
            if self._dp_gate():
                self.model.train_frames(agents, seqs, r_, a_)
                self._count_train_step(agents.shape[0], None, r_, a_, frames=(agents, seqs))

    def save_model(self):
        self.model.save(self.stats.episode_count.value)

    # ---- main loop (Server.py:155-198) ----------------------------------------------------------
    def main(self, max_seconds=None):
        self.stats.start()
        self.dynamic_adjustment.start()
        lr_mult = (Config.LEARNING_RATE_END - Config.LEARNING_RATE_START) / Config.ANNEALING_EPISODE_COUNT
        beta_mult = (Config.BETA_END - Config.BETA_START) / Config.ANNEALING_EPISODE_COUNT
        t0 = time.time()
        try:
            while self.dp is not None or self.stats.episode_count.value < Config.EPISODES:
                step = min(self.stats.episode_count.value, Config.ANNEALING_EPISODE_COUNT - 1)
                lr = Config.LEARNING_RATE_START + lr_mult * step
                beta = Config.BETA_START + beta_mult * step
                if Config.SAVE_MODELS and self.stats.should_save_model.value > 0:
                    self.save_model()
                    self.stats.should_save_model.value = 0
                if self.failure is not None:
                    break
                timed_out = max_seconds is not None and time.time() - t0 > max_seconds
                if self.dp is not None:
                    # rank 0 decides when to stop and which lr / beta each step uses; every rank leaves on the same step
                    want_stop = timed_out or self.stats.episode_count.value >= Config.EPISODES
                    try:
                        self.dp.poll(want_stop, self.training_step, lr, beta)
                    except DataParallel.GroupStalled as e:       # a rank keeps the others inside a collective: give up, say who
                        self.worker_failed("data-parallel group", e)
                        break
                    if self.dp.finished(self.training_step):
                        break
                else:
                    self.model.learning_rate, self.model.beta = lr, beta
                    if timed_out:
                        break
                time.sleep(0.01)
        finally:
            self.shutdown()
        if self.failure is not None:
            raise RuntimeError("%s died: %r" % self.failure) from self.failure[1]

    def shutdown(self):
        self.closing = True
        self.dynamic_adjustment.exit_flag = True
        if self.dynamic_adjustment.is_alive():      # it may still be starting workers (a run that ends at once)
            self.dynamic_adjustment.join(timeout=30)
        for a in self.agents:
            a.exit_flag.value = True
        self.transport.shutdown()
        while self.agents:
            self.remove_agent()
        while self.predictors:
            self.remove_predictor()
        while self.trainers:
            self.remove_trainer()
        if self.stats.is_alive():
            self.stats.terminate()
        stalled = self.failure is not None and isinstance(self.failure[1], DataParallel.GroupStalled)
        if self.zero_copy and not stalled:          # (a stalled group: the train stream sits in a collective that never
            self.model.unregister_transport()       # completes and a device sync would hang -- the caller ends the process)
            self.zero_copy = False                  # unpin before the segment is unmapped
        if not stalled:
            self.transport.close()
        else:
            self.transport.unlink()                 # no unmap under a GPU that may still read it, but no /dev/shm leftover either

    @staticmethod
    def get_state_dim():
        return Environment.get_state_dim()

    @staticmethod
    def get_num_action():
        if Config.FRAME_SOURCE == 'gym':       # as the reference asks a throw-away Environment (Server.py:200-202);
            Environment()                      # it also fixes NUM_ACTIONS and the frame size in Config
        return int(Config.NUM_ACTIONS)
