"""Starts the initial trainers / predictors / agents and, when DYNAMIC_SETTINGS is on, tunes their
counts by a random walk that keeps a move only if the number of train steps in the following
window did not drop (reference: ga3c/ThreadDynamicAdjustment.py:35-144).

One deliberate difference: the reference's branch that REMOVES agents
repeats the `<` condition of the add branch (:75,:79) and can never run; the evident intent (`>`)
is implemented here.  Human-reference agents (:84-93) are pyperrace-only and out of scope.
"""
from threading import Event, Thread

import numpy as np

from Config import Config


class ThreadDynamicAdjustment(Thread):
    def __init__(self, server):
        super(ThreadDynamicAdjustment, self).__init__()
        self.daemon = True
        self.server = server
        self.enabled = Config.DYNAMIC_SETTINGS
        self.trainer_count = Config.TRAINERS
        self.predictor_count = Config.PREDICTORS
        self.agent_count = Config.AGENTS
        self.temporal_training_count = 0
        self._halt = Event()            # set through exit_flag; the waits below end at once

    @property
    def exit_flag(self):
        return self._halt.is_set()

    @exit_flag.setter
    def exit_flag(self, value):
        if value:
            self._halt.set()
        else:
            self._halt.clear()

    @staticmethod
    def _resize(current, wanted, add, remove):
        for _ in range(current, wanted):
            add()
        for _ in range(wanted, current):
            remove()

    def enable_disable_components(self):
        s = self.server
        self._resize(len(s.trainers), self.trainer_count, s.add_trainer, s.remove_trainer)
        self._resize(len(s.predictors), self.predictor_count, s.add_predictor, s.remove_predictor)
        self._resize(len(s.agents), self.agent_count, s.add_agent, s.remove_agent)
        # an add can be refused (no agent slot left): the walk, the accept / revert decision and the status line go on from
        # what is really running, not from a move that did not happen
        self.trainer_count, self.predictor_count, self.agent_count = len(s.trainers), len(s.predictors), len(s.agents)

    def random_walk(self):
        # one of {-1, 0, +1} for each of trainers, predictors, agents; never below 1
        direction = np.random.randint(3, size=3) - 1
        self.trainer_count = max(1, self.trainer_count - direction[0])
        self.predictor_count = max(1, self.predictor_count - direction[1])
        self.agent_count = min(max(1, self.agent_count - direction[2]), self.server.max_agents)

    def update_stats(self):
        st = self.server.stats
        st.trainer_count.value = self.trainer_count
        st.predictor_count.value = self.predictor_count
        st.agent_count.value = self.agent_count

    def run(self):
        self.enable_disable_components()
        self.update_stats()
        if not self.enabled:
            return
        self._halt.wait(Config.DYNAMIC_SETTINGS_INITIAL_WAIT)
        while not self.exit_flag:
            before = (self.trainer_count, self.predictor_count, self.agent_count)
            self.random_walk()
            if (self.trainer_count, self.predictor_count, self.agent_count) == before:
                continue
            old_count = self.temporal_training_count
            self.enable_disable_components()
            self.temporal_training_count = 0
            if self._halt.wait(Config.DYNAMIC_SETTINGS_STEP_WAIT):
                break
            if self.temporal_training_count < old_count:          # it got worse: go back
                self.trainer_count, self.predictor_count, self.agent_count = before
            self.update_stats()
