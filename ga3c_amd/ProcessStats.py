"""Statistics process: consumes (time, reward, length) per finished episode, appends results.txt,
prints the status line, raises should_save_model.  Wire format, status-line format and the
definitions of PPS / TPS are the reference's (ga3c/ProcessStats.py:42-108):
  PPS = frames of FINISHED episodes / seconds since start, TPS = training_count / seconds, both ceil'ed.
"""
import sys
import time
from collections import deque
from datetime import datetime

import numpy as np

from Config import Config
from ProcessAgent import MP


class ProcessStats(MP.Process):
    def __init__(self):
        super(ProcessStats, self).__init__()
        self.daemon = True
        self.episode_log_q = MP.Queue(maxsize=100)
        self.episode_count = MP.Value('i', 0)
        self.training_count = MP.Value('i', 0)
        self.should_save_model = MP.Value('i', 0)
        self.trainer_count = MP.Value('i', 0)
        self.predictor_count = MP.Value('i', 0)
        self.agent_count = MP.Value('i', 0)
        self.replay_memory_size = MP.Value('i', 0)
        self.total_frame_count = 0
        self.start_time = time.time()
        self.config = {k: v for k, v in vars(Config).items() if k.isupper()}

    def FPS(self):
        return np.ceil(self.total_frame_count / (time.time() - self.start_time))

    def TPS(self):
        return np.ceil(self.training_count.value / (time.time() - self.start_time))

    @staticmethod
    def status_line(elapsed, episode, reward, rscore, rpps, pps, tps, nt, np_, na, rsize):
        return ('[Time: %8d] [Episode: %8d Score: %10.4f] [RScore: %10.4f RPPS: %5d] [PPS: %5d TPS: %5d] '
                '[NT: %2d NP: %2d NA: %2d][RSize: %8d]'
                % (elapsed, episode, reward, rscore, rpps, pps, tps, nt, np_, na, rsize))

    def run(self):
        for k, v in self.config.items():
            setattr(Config, k, v)
        window = deque()
        rolling_frames, rolling_reward = 0, 0
        self.start_time = time.time()
        first_time = datetime.now()
        with open(Config.RESULTS_FILENAME, 'a') as results_logger:
            while True:
                episode_time, reward, length = self.episode_log_q.get()
                results_logger.write('%s, %d, %d\n' % (episode_time.strftime("%Y-%m-%d %H:%M:%S"), reward, length))
                results_logger.flush()
                self.total_frame_count += length
                self.episode_count.value += 1
                rolling_frames += length
                rolling_reward += reward
                if len(window) == Config.STAT_ROLLING_MEAN_WINDOW:
                    old_time, old_reward, old_length = window.popleft()
                    rolling_frames -= old_length
                    rolling_reward -= old_reward
                    first_time = old_time
                window.append((episode_time, reward, length))
                if self.episode_count.value % Config.SAVE_FREQUENCY == 0:
                    self.should_save_model.value = 1
                if self.episode_count.value % Config.PRINT_STATS_FREQUENCY == 0:
                    span = max((datetime.now() - first_time).total_seconds(), 1e-9)
                    print(self.status_line(int(time.time() - self.start_time), self.episode_count.value, reward,
                                           rolling_reward / len(window), rolling_frames / span, self.FPS(), self.TPS(),
                                           self.trainer_count.value, self.predictor_count.value,
                                           self.agent_count.value, self.replay_memory_size.value))
                    sys.stdout.flush()
