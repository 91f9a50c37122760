"""One agent step, as ProcessAgent records it (reference: ga3c/Experience.py:27-34)."""


class Experience:
    __slots__ = ("state", "action", "prediction", "reward", "next_state", "done")

    def __init__(self, state, action, prediction, reward, next_state, done):
        self.state, self.action, self.prediction = state, action, prediction
        self.reward, self.next_state, self.done = reward, next_state, done
