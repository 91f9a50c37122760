"""Frame source with the reference Atari wrapper's interface and state layout
(ga3c/Environment.py:41-93): reset(), step(action) -> (reward, done), .previous_state,
.current_state = f32 [84,84,4] HWC with values k/128 - 1, get_num_actions().

gym / ALE are absent offline, so the default source is synthetic: every step
shifts one new 84x84 uint8 plane, k ~ U{0..255} from PCG64(RANDOM_SEED + agent id), into a 4-deep FIFO;
episodes last SYNTHETIC_EPISODE_LENGTH steps; reward is +-1 with probability 0.01 each; `done` on the
last step.  The uint8 stack is exposed as .current_u8 / .previous_u8 so the transport can ship 28,224
bytes per state; the f32 views are computed on demand with the reference's own arithmetic.
"""
import numpy as np

from Config import Config


def u8_to_f32(frames):
    return frames.astype(np.float32) / np.float32(128.0) - np.float32(1.0)     # Environment.py:60


class Environment:
    def __init__(self, agent_id=0):
        self.nb_frames = Config.STACKED_FRAMES
        self.rng = np.random.Generator(np.random.PCG64(Config.RANDOM_SEED + int(agent_id)))
        self.num_actions = int(Config.NUM_ACTIONS)
        self.episode_length = int(Config.SYNTHETIC_EPISODE_LENGTH)
        self.previous_u8 = None
        self.current_u8 = None
        self.total_reward = 0
        self.reset()

    # ---- reference interface
    def get_num_actions(self):
        return self.num_actions

    @staticmethod
    def get_state_dim():
        return (Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH, Config.STACKED_FRAMES)

    @property
    def current_state(self):
        return None if self.current_u8 is None else u8_to_f32(self.current_u8)

    @property
    def previous_state(self):
        return None if self.previous_u8 is None else u8_to_f32(self.previous_u8)

    def reset(self):
        self.total_reward = 0
        self._t = 0
        self._frames = []
        self._push_frame()
        self.previous_u8 = self.current_u8 = None

    def step(self, action):
        self._t += 1
        draw = self.rng.random()
        reward = 1.0 if draw < 0.01 else (-1.0 if draw < 0.02 else 0.0)
        done = self._t >= self.episode_length + self.nb_frames - 1
        self.total_reward += reward
        self._push_frame()
        self.previous_u8 = self.current_u8
        self.current_u8 = self._stack()
        return reward, done

    # ---- internals
    def _push_frame(self):
        if len(self._frames) == self.nb_frames:
            self._frames.pop(0)
        self._frames.append(self.rng.integers(0, 256, size=(Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH), dtype=np.uint8))

    def _stack(self):
        if len(self._frames) < self.nb_frames:
            return None                                   # frame queue not full yet (Environment.py:64-65)
        return np.ascontiguousarray(np.stack(self._frames, axis=-1))     # [84,84,4] HWC (Environment.py:66-68)
