"""End-to-end plumbing on CPU: real agent processes (forkserver), the shared-memory transport, the
predictor / trainer threads, the stats process and the Server loop -- with a stand-in model so that no
GPU is needed.  Checks the rollout shapes the reference produces (first rollout TIME_MAX+1 rows, later
ones TIME_MAX + 1 carried) and the results.txt wire format."""
import os
import re

import numpy as np
import pytest


class _StandInModel:
    """Uniform policy, zero value; records what the batching threads hand over."""
    def __init__(self, n_act):
        self.n_act = n_act
        self.learning_rate = self.beta = 0.0
        self.pred_batches, self.train_rows, self.train_dtypes = [], [], set()

    def predict_p_and_v(self, x):
        self.pred_batches.append(x.shape)
        b = x.shape[0]
        return np.full((b, self.n_act), 1.0 / self.n_act, np.float32), np.zeros(b, np.float32)

    def train(self, x, y_r, a, x2, done, tid):
        assert x.shape[1:] == (84, 84, 4) and a.shape == (x.shape[0], self.n_act) and y_r.shape == (x.shape[0],)
        assert np.all(a.sum(axis=1) == 1.0)
        self.train_rows.append(x.shape[0])
        self.train_dtypes.add(str(x.dtype))

    def save(self, episode):
        pass

    def log(self, *a, **k):
        pass


@pytest.mark.timeout(120)
def test_server_runs_agents_predictor_trainer_stats(tmp_path, monkeypatch):
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    saved = {k: getattr(Config, k) for k in ("AGENTS", "PREDICTORS", "TRAINERS", "SYNTHETIC_EPISODE_LENGTH", "TIME_MAX",
                                             "DYNAMIC_SETTINGS", "SAVE_MODELS", "TRAINING_MIN_BATCH_SIZE", "NUM_ACTIONS")}
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 3, 1, 1
    Config.SYNTHETIC_EPISODE_LENGTH, Config.TIME_MAX = 23, 5
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS = False, False
    Config.TRAINING_MIN_BATCH_SIZE, Config.NUM_ACTIONS = 0, 6
    try:
        from Server import Server
        model = _StandInModel(6)
        srv = Server(model=model, max_agents=8)
        srv.main(max_seconds=6)
        assert srv.predictions_served > 50
        assert model.train_rows and model.train_dtypes == {"uint8"}
        # 23-step episodes cut at TIME_MAX=5: rollouts of 6 rows (first: TIME_MAX+1; later: 1 carried + 5), last one shorter
        assert max(model.train_rows) == 6 and set(model.train_rows) <= {1, 2, 3, 4, 5, 6}
        assert all(s[1:] == (84, 84, 4) for s in model.pred_batches)
        assert srv.training_step == len(model.train_rows) == srv.stats.training_count.value
        lines = open("results.txt").read().strip().splitlines()
        assert lines and all(re.match(r"^\d{4}-\d\d-\d\d \d\d:\d\d:\d\d, -?\d+, \d+$", ln) for ln in lines)
        # frame accounting of ProcessAgent.py:174: each rollout contributes len(r_)+1 and carries one row over:
        # 23 steps -> rollouts of 6,6,6,6,3 rows -> 7+7+7+7+4 = 32
        lengths = [int(ln.split(", ")[2]) for ln in lines]
        assert set(lengths) == {32}
    finally:
        for k, v in saved.items():
            setattr(Config, k, v)


def test_status_line_format_is_the_references():
    import ga3c_amd  # noqa: F401
    from ProcessStats import ProcessStats
    line = ProcessStats.status_line(35, 30, -20.0, -20.4, 899, 900, 186, 2, 2, 32, 0)
    assert line == ("[Time:       35] [Episode:       30 Score:   -20.0000] [RScore:   -20.4000 RPPS:   899] "
                    "[PPS:   900 TPS:   186] [NT:  2 NP:  2 NA: 32][RSize:        0]")
