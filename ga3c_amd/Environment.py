"""Frame source with the reference Atari wrapper's interface and state layout
(ga3c/Environment.py:41-93): reset(), step(action) -> (reward, done), .previous_state,
.current_state = f32 [84,84,4] HWC with values k/128 - 1, get_num_actions().

gym / ALE are absent offline, so the default source is synthetic: every step
shifts one new 84x84 uint8 plane, k ~ U{0..255} from PCG64(RANDOM_SEED + agent id), into a 4-deep FIFO;
episodes last SYNTHETIC_EPISODE_LENGTH steps; reward is +-1 with probability 0.01 each; `done` on the
last step.  The uint8 stack is exposed as .current_u8 / .previous_u8 so the transport can ship 28,224
bytes per state; the f32 views are computed on demand with the reference's own arithmetic.

The FIFO is kept as one little-endian uint32 per pixel (byte c = frame c, oldest first): pushing a frame is
`(stack >> 8) | (frame << 24)` (ga3c_frame_queue_push in libga3c_host.so), which yields the same [84,84,4] bytes as
np.stack(frames, axis=-1) in a tenth of the time, and Generator.bytes() yields the same stream as integers(0, 256, dtype=uint8)
(tests/test_control_plane_cpu.py pins both equalities).
"""
import sys

import numpy as np

from Config import Config


def u8_to_f32(frames):
    return frames.astype(np.float32) / np.float32(128.0) - np.float32(1.0)     # Environment.py:60


class Environment:
    """Config.FRAME_SOURCE = 'rgb' swaps the 84x84 plane source for a stand-in emulator: every episode draws one random
    FRAME_HEIGHT x FRAME_WIDTH x 3 background and every step moves an 8x8 sprite of a fresh colour over it and redraws
    one row of noise -- cheap, deterministic per seed, and enough structure for the contrast stretch and the resize to
    matter.  FRONTEND = 'host' then runs the reference's _preprocess here (ga3c_frame_preprocess); FRONTEND = 'device'
    only exposes the raw frame (.frame) and the number of frames since reset (.frames_queued): the state lives in HBM."""

    def __init__(self, agent_id=0):
        self.gym = None
        if Config.FRAME_SOURCE == 'gym':
            # Optional real emulator (Environment.py:41-50, GameManager.py:30-49): gym.make(Config.GAME) supplies the RGB
            # frames, rewards and `done`; everything downstream (front-end, frame queue, transport) is the 'rgb' path.
            # gym / ALE are not part of this image, so this branch is untested offline and fails loudly without them.
            try:
                import gym
            except ImportError as e:
                raise ImportError("Config.FRAME_SOURCE = 'gym' needs the gym package with Atari support (%s); the offline "
                                  "sources are 'planes' and 'rgb'" % e)
            self.gym = gym.make(Config.GAME)
            Config.NUM_ACTIONS = int(self.gym.action_space.n)
            shape = self.gym.observation_space.shape
            Config.FRAME_HEIGHT, Config.FRAME_WIDTH = int(shape[0]), int(shape[1])
        self.rgb = Config.FRAME_SOURCE in ('rgb', 'gym')
        self.on_device = Config.FRONTEND == 'device'      # the frame queue lives in HBM: .frame is all the agent holds
        self.frame = None
        self.frames_queued = 0
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
        self.frames_queued = 0
        self._filled = 0
        self._stack32 = np.zeros((Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH), np.uint32)
        self._stack_addr = self._stack32.ctypes.data
        if self.gym is not None:
            obs = self.gym.reset()
            self._gym_frame = np.ascontiguousarray(obs[0] if isinstance(obs, tuple) else obs, dtype=np.uint8)
        self._push_frame()
        self.previous_u8 = self.current_u8 = None

    def step(self, action):
        self._t += 1
        if self.gym is not None:
            res = self.gym.step(int(action))               # (obs, reward, done, info) or (obs, reward, term, trunc, info)
            self._gym_frame = np.ascontiguousarray(res[0], dtype=np.uint8)
            reward, done = float(res[1]), bool(res[2]) or (len(res) == 5 and bool(res[3]))
        else:
            draw = self.rng.random()
            reward = 1.0 if draw < 0.01 else (-1.0 if draw < 0.02 else 0.0)
            done = self._t >= self.episode_length + self.nb_frames - 1
        self.total_reward += reward
        self._push_frame()
        self.previous_u8 = self.current_u8
        self.current_u8 = self._stack()
        return reward, done

    # ---- internals
    def _emulate(self):
        """The stand-in emulator's next RGB frame."""
        if self.gym is not None:
            return self._gym_frame
        fh, fw = Config.FRAME_HEIGHT, Config.FRAME_WIDTH
        if self.frames_queued == 0:
            self._background = np.frombuffer(self.rng.bytes(fh * fw * 3), np.uint8).reshape(fh, fw, 3)
        frame = self._background.copy()
        draw = np.frombuffer(self.rng.bytes(8 + fw * 3), np.uint8)
        y = int(draw[0]) * (fh - 8) // 255
        x = int(draw[1]) * (fw - 8) // 255
        frame[y:y + 8, x:x + 8] = draw[2:5]
        frame[int(draw[5]) * (fh - 1) // 255] = draw[8:].reshape(fw, 3)
        return frame

    def _push_frame(self):
        h, w = Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH
        plane_arg = None
        if self.rgb:
            self.frame = self._emulate()
            self.frames_queued += 1
            if self.on_device:
                return
            import _native as nat
            frame = np.empty((h, w), np.uint8)
            fh, fw, fc = self.frame.shape
            nat.check_host(nat.host_lib().ga3c_frame_preprocess(nat.ptr(self.frame, nat.u8p), fh, fw, fc, h, w,
                                                                nat.ptr(frame, nat.u8p)), "ga3c_frame_preprocess")
        else:
            if (h * w) % 8 == 0:
                # the same byte stream as rng.bytes(h * w) -- both hand out PCG64's 64-bit outputs low half first -- drawn
                # as whole words in one vectorised call: 5 us instead of 13 (equality pinned in tests/test_control_plane_cpu.py)
                raw = self.rng.bit_generator.random_raw(h * w // 8)
                plane_arg = raw.ctypes.data
                frame = raw.view(np.uint8).reshape(h, w) if self.on_device else None
                self._keep = raw                        # the plane's bytes must outlive the push below
            else:
                plane_arg = self.rng.bytes(h * w)       # (bytes go to C as they are: no address lookup)
                frame = np.frombuffer(plane_arg, np.uint8).reshape(h, w) if self.on_device else None
            if self.on_device:                            # ready-made plane, queue kept on the device
                self.frame = frame
                self.frames_queued += 1
                return
        # (stack >> 8) | (frame << 24) into a NEW array (the experiences of the running rollout keep the old states alive)
        nxt = np.empty((h, w), np.uint32)
        addr = nxt.ctypes.data                      # (one address lookup per step: the old array's is remembered)
        _push(self._stack_addr, plane_arg if plane_arg is not None else frame.ctypes.data, addr, h * w)
        self._stack32, self._stack_addr = nxt, addr
        self._filled = min(self._filled + 1, self.nb_frames)

    def _stack(self):
        if self._filled < self.nb_frames:
            return None                                   # frame queue not full yet (Environment.py:64-65)
        return self._stack32.view(np.uint8).reshape(Config.IMAGE_HEIGHT, Config.IMAGE_WIDTH, 4)   # HWC (:66-68)


assert sys.byteorder == "little" and Config.STACKED_FRAMES == 4, "the uint32 frame FIFO assumes 4 frames, LE bytes"
import _native as _nat    # noqa: E402

_push = _nat.host_lib().ga3c_frame_queue_push      # the FIFO push in C (include/ga3c_host.h): 6.8 -> ~2 us per agent step
