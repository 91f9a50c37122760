"""Actor process: environment loop, predict() round trip, action sampling, rollout cut, returns,
hand-off to the trainer.  Behaviour follows the reference's ProcessAgent
(/root/reference/ga3c/ProcessAgent.py:44-178); the three pickling queues it talks to are replaced by
the shared-memory transport (Transport.py / include/ga3c_host.h):

  predict()            :102-107  state -> own slot, id -> request ring, sleep on the slot's futex
  _accumulate_rewards  :69-84    bit-exact, computed by ga3c_returns_fork (f64, sequential products)
  rollout cut          :145      on done or time_count == TIME_MAX; last experience re-used (:159)
  run()                :164-178  rollout -> training slot; episode totals -> episode_log_q

With Config.FRAME_SOURCE = 'rgb' and FRONTEND = 'device' the agent never holds a state: it writes the emulator's raw
frame into its slot, the predictor pushes it into the agent's device-side frame queue and answers with the prediction
for the state that push completed, and an experience names its state by the sequence number of that plane
(run_episode_device, _ship; include/ga3c_abi.h: ga3c_net_frames_push_offsets / ga3c_net_train_frames).

The debug prints of :131,133,152 are not reproduced; x2_/done_ stay in
convert_data's signature but are not transported (Q10).
"""
import multiprocessing as mp
import os
import time
from datetime import datetime

import numpy as np

from Config import Config
from Environment import Environment, u8_to_f32
from Experience import Experience
import Transport as tp

MP = mp.get_context("forkserver")     # children never inherit the server's HIP state
# the fork server imports the agent's modules once; every agent then starts as a fork of that warm process
MP.set_forkserver_preload(["numpy", "Config", "Experience", "Environment", "_native", "Transport", "ProcessAgent"])


def config_snapshot():
    return {k: v for k, v in vars(Config).items() if k.isupper()}


class ProcessAgent(MP.Process):
    def __init__(self, id, transport_name, episode_log_q, config=None, planes_pushed=0):
        super(ProcessAgent, self).__init__()
        self.id = id
        self.transport_name = transport_name
        self.episode_log_q = episode_log_q
        self.config = config if config is not None else config_snapshot()
        self.discount_factor = self.config["DISCOUNT"]
        self.exit_flag = MP.Value('i', 0)
        self.time_count = 0
        self.transport = None
        self.env = None
        # device front-end: frames handed over under this id so far = sequence number of the next plane (not 0 when the
        # id was another agent's before: the device keeps counting, Server.add_agent)
        self.planes_pushed = int(planes_pushed)
        self.requests = 0                # number of this id's newest request (run() reads where the id's counter stands)
        self.names_states = False        # STATE_CACHE_ACTIVE: experiences name their state by that number

    # ---- pieces with the reference's names and semantics ------------------------------------
    @staticmethod
    def _accumulate_rewards(experiences, discount_factor, terminal_reward):
        if Config.RETURN_MODE == 'nstep':
            # upstream GA3C: R = clip(r) + gamma R from the bootstrap value, last row dropped
            out = tp.returns_nstep([e.reward for e in experiences], discount_factor, terminal_reward,
                                   Config.REWARD_MIN, Config.REWARD_MAX)
            for e, r in zip(experiences, out):
                e.reward = float(r)
            return experiences[:-1]
        if (not Config.REWARD_CLIPPING and Config.DISCOUNTING and Config.USE_INTERMEDIATE_REWARD
                and len(experiences) > 1):
            # the reference reads the clipped reward it never computed (ProcessAgent.py:73-80); SURVEY section 9, Q2
            raise UnboundLocalError("local variable 'r' referenced before assignment")
        out = tp.accumulate_rewards_fork([e.reward for e in experiences], discount_factor, terminal_reward,
                                         Config.DISCOUNTING, Config.USE_INTERMEDIATE_REWARD)
        for e, r in zip(experiences, out):
            e.reward = float(r)
        return experiences                      # all T rows (ProcessAgent.py:84)

    def convert_data(self, experiences):
        x_ = np.array([u8_to_f32(e.state) if e.state.dtype == np.uint8 else e.state for e in experiences])
        x2_ = np.array([u8_to_f32(e.next_state) if e.next_state.dtype == np.uint8 else e.next_state
                        for e in experiences])
        done_ = np.array([e.done for e in experiences])
        a_ = np.eye(self.num_actions)[np.array([e.action for e in experiences])].astype(np.float32)
        r_ = np.array([e.reward for e in experiences])
        return x_, r_, a_, x2_, done_

    def predict(self, state):
        """state: uint8 [84,84,4] frames (STATE_TRANSPORT='u8') or f32 [84,84,4]."""
        slot = self.transport.state_view(self.id, state.dtype)
        slot[:] = state.reshape(-1)
        if self.transport.submit(self.id) == tp.CLOSED:     # nothing was queued: waiting would return the previous answer
            raise SystemExit(0)
        self.requests += 1                                  # = this request's number (ga3c_pq_request_seq)
        while True:
            rc, p, v = self.transport.wait(self.id, Config.QUEUE_TIMEOUT_MS)
            if rc == 0:
                return p, v
            if rc == tp.CLOSED or self.exit_flag.value:
                raise SystemExit(0)

    def predict_and_select(self, state, flags=0):
        """predict() + select_action() of one step in ONE foreign call (ga3c_pq_round_trip): state (or raw frame) into the
        slot, submit, wait, draw.  The uniform comes from the global RandomState exactly where np.random.choice would draw
        it -- one draw per step, none in PLAY_MODE (ProcessAgent.py:102-115) -- so seeds give the reference's actions.
        -> (prediction, value, action)."""
        u = -1.0 if Config.PLAY_MODE else np.random.random_sample()
        flat = state.reshape(-1)
        rc, p, v, a = self.transport.round_trip(self.id, flat if flat.flags.c_contiguous else np.ascontiguousarray(flat),
                                                flags, Config.QUEUE_TIMEOUT_MS, u)
        if rc == tp.CLOSED:                                 # nothing was queued
            raise SystemExit(0)
        while rc != 0:
            if rc == tp.CLOSED or self.exit_flag.value:
                raise SystemExit(0)
            rc, p, v, a = self.transport.round_trip(self.id, None, flags, Config.QUEUE_TIMEOUT_MS, u, submit=False)
        return p, v, int(self.actions[a]) if a >= 0 else int(np.argmax(p))

    def push_frame(self, frame, flags):
        """Device front-end: the raw frame goes into this agent's slot; the answer is (p, v) of the state the frame
        completed (meaningless when flags ask for no prediction)."""
        n = frame.size
        self.transport.state_view(self.id)[:n] = frame.reshape(-1)
        if self.transport.submit(self.id, flags) == tp.CLOSED:     # nothing was queued (see predict)
            raise SystemExit(0)
        self.planes_pushed += 1
        while True:
            rc, p, v = self.transport.wait(self.id, Config.QUEUE_TIMEOUT_MS)
            if rc == 0:
                return p, v
            if rc == tp.CLOSED or self.exit_flag.value:
                raise SystemExit(0)

    @staticmethod
    def select_action(actions, prediction):
        """Reference: np.random.choice(actions, p=prediction) (ProcessAgent.py:109-115).  This is that call's own
        algorithm -- normalised float64 cdf, one uniform from the global RandomState, searchsorted(side='right') --
        without its per-call argument validation (~30 us); same seed, same draws (tests/test_control_plane_cpu.py).
        For the float32 vector a prediction is, the cdf and the search run in C (ga3c_select_action: 5.9 -> 4.4 us)."""
        if Config.PLAY_MODE:
            return int(np.argmax(prediction))
        if prediction.dtype == np.float32 and prediction.flags.c_contiguous and prediction.size <= 64:
            idx = tp.select_action_index(prediction, np.random.random_sample())
            return int(actions[idx])
        cdf = np.cumsum(prediction, dtype=np.float64)
        cdf /= cdf[-1]
        return int(actions[min(int(cdf.searchsorted(np.random.random_sample(), side='right')), len(cdf) - 1)])

    # ---- episode loop ------------------------------------------------------------------------
    def run_episode(self):
        self.env.reset()
        done = False
        experiences = []
        self.time_count = 0
        reward_sum = 0.0
        as_u8 = Config.STATE_TRANSPORT == 'u8'
        while not done:
            if self.env.current_u8 is None:
                self.env.step(None)             # frame queue still filling (ProcessAgent.py:127-129)
                continue
            state = self.env.current_u8 if as_u8 else self.env.current_state
            prediction, value, action = self.predict_and_select(state)
            self.requests += 1                                  # = this request's number (ga3c_pq_request_seq)
            reward, done = self.env.step(action)
            reward_sum += reward
            if self.names_states:               # the engine kept the state this request carried: the experience names it
                experiences.append(Experience(self.requests, action, prediction, reward, None, done))
            else:
                next_state = self.env.current_u8 if as_u8 else self.env.current_state
                experiences.append(Experience(state, action, prediction, reward, next_state, done))
            if done or self.time_count == Config.TIME_MAX:
                if Config.RETURN_MODE == 'nstep':
                    terminal_reward = 0.0 if done else float(value)
                else:
                    terminal_reward = reward    # the fork's choice (ProcessAgent.py:148)
                updated = ProcessAgent._accumulate_rewards(experiences, self.discount_factor, terminal_reward)
                yield updated, reward_sum
                self.time_count = 0
                experiences = [experiences[-1]]
                reward_sum = 0.0
            self.time_count += 1

    def run_episode_device(self):
        """run_episode with the frame queue on the device: one round trip per emulator step hands the newest frame over
        and brings back the prediction for the state it completes (same control flow as ProcessAgent.py:117-162)."""
        env = self.env
        env.reset()
        done = False
        experiences = []
        self.time_count = 0
        reward_sum = 0.0
        flags = tp.REQ_RESET
        while not done:
            full = env.frames_queued >= env.nb_frames
            if not full:                        # frame queue still filling (ProcessAgent.py:127-129): push only, no draw
                self.push_frame(env.frame, flags | tp.REQ_NO_PREDICT)
                flags = 0
                env.step(None)
                continue
            prediction, value, action = self.predict_and_select(env.frame, flags)
            self.planes_pushed += 1
            flags = 0
            state = self.planes_pushed - 1      # the state is named by its newest plane
            reward, done = env.step(action)
            reward_sum += reward
            experiences.append(Experience(state, action, prediction, reward, None, done))
            if done or self.time_count == Config.TIME_MAX:
                terminal_reward = (0.0 if done else float(value)) if Config.RETURN_MODE == 'nstep' else reward
                updated = ProcessAgent._accumulate_rewards(experiences, self.discount_factor, terminal_reward)
                yield updated, reward_sum
                self.time_count = 0
                experiences = [experiences[-1]]
                reward_sum = 0.0
            self.time_count += 1

    def _ship(self, experiences):
        """Rollout -> one slot of the training queue (stands for training_q.put, ProcessAgent.py:175)."""
        while True:
            slot = self.transport.acquire(Config.QUEUE_TIMEOUT_MS)
            if slot >= 0:
                break
            if slot == tp.CLOSED or self.exit_flag.value:
                raise SystemExit(0)
        states, returns, actions = self.transport.rollout_views(slot)
        n = len(experiences)
        for i, e in enumerate(experiences):
            if self.env.on_device or self.names_states:   # row = (plane / request number, agent id): the state itself is in HBM
                states[i, :8].view(np.int64)[0] = e.state
                states[i, 8:12].view(np.int32)[0] = self.id
            else:
                states[i] = e.state.reshape(-1).view(np.uint8)
            returns[i] = e.reward               # f64 -> f32 here, as TF's feed does (NetworkVP.py:70,256)
            actions[i] = e.action
        self.transport.commit(slot, n)

    def run(self):
        for k, v in self.config.items():
            setattr(Config, k, v)
        cpus = getattr(Config, "AGENT_CPUS", None)            # Placement.py: the server's batching threads keep CPUs of their own
        if cpus:
            try:
                os.sched_setaffinity(0, cpus)
            except OSError:
                pass
        self.transport = tp.Transport.attach(self.transport_name)
        self.names_states = bool(getattr(Config, "STATE_CACHE_ACTIVE", False))
        self.requests = self.transport.request_seq(self.id)   # (not 0 when the id was another agent's before)
        self.env = Environment(self.id)
        self.num_actions = self.env.get_num_actions()
        self.actions = np.arange(self.num_actions)
        time.sleep(np.random.rand() * 0.2)                              # staggered start (:167)
        np.random.seed(np.int32(time.time() % 1 * 1000 + self.id * 10))  # (:168)
        try:
            while self.exit_flag.value == 0:
                total_reward = 0
                total_length = 0
                finished = True
                episode = self.run_episode_device if self.env.on_device else self.run_episode
                for experiences, reward_sum in episode():
                    total_reward += reward_sum
                    total_length += len(experiences) + 1        # frame accounting of :174
                    if experiences:
                        self._ship(experiences)
                    if self.exit_flag.value and not (experiences and experiences[-1].done):
                        finished = False                        # asked to stop mid-episode: log nothing
                        break
                if finished:
                    self.episode_log_q.put((datetime.now(), total_reward, total_length))
        except SystemExit:
            pass
        finally:
            self.transport.close()
