"""-m gpu: the whole actor-learner engine on the real HIP Network -- agent processes, shared-memory
transport, predictor and trainer threads -- for a few seconds; then the weights must have moved and
every prediction must be a probability vector."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(180)
def test_engine_trains_on_gpu(tmp_path, monkeypatch):
    import ga3c_amd  # noqa: F401
    from Config import Config
    monkeypatch.chdir(tmp_path)
    saved = {k: getattr(Config, k) for k in ("AGENTS", "PREDICTORS", "TRAINERS", "SYNTHETIC_EPISODE_LENGTH", "TIME_MAX",
                                             "DYNAMIC_SETTINGS", "SAVE_MODELS", "TRAINING_MIN_BATCH_SIZE", "NUM_ACTIONS",
                                             "PREDICTION_BATCH_SIZE")}
    Config.AGENTS, Config.PREDICTORS, Config.TRAINERS = 6, 2, 1
    Config.SYNTHETIC_EPISODE_LENGTH, Config.TIME_MAX = 40, 5
    Config.DYNAMIC_SETTINGS, Config.SAVE_MODELS = False, True
    Config.TRAINING_MIN_BATCH_SIZE, Config.NUM_ACTIONS, Config.PREDICTION_BATCH_SIZE = 11, 6, 32
    try:
        from Server import Server
        srv = Server(max_agents=8)
        before = srv.model.get_arena(0)
        srv.main(max_seconds=5)
        after = srv.model.get_arena(0)
        assert srv.predictions_served > 100 and srv.training_step > 5
        assert srv.model.get_global_step() == srv.training_step
        assert np.all(np.isfinite(after)) and np.max(np.abs(after - before)) > 1e-5
        # checkpoint round trip through the reference's naming scheme (checkpoints/<name>_%08d)
        srv.model.save(7)
        srv.model.set_arena(0, np.zeros_like(after))
        Config.LOAD_EPISODE = 7          # the run itself may have saved later episodes (SAVE_FREQUENCY)
        assert srv.model.load() == 7
        Config.LOAD_EPISODE = 0
        assert np.array_equal(srv.model.get_arena(0), after)
        p, v = srv.model.predict_p_and_v(np.zeros((3, 84, 84, 4), np.float32))
        assert np.allclose(p.sum(axis=1), 1.0, atol=1e-5)
        srv.model.close()
    finally:
        for k, v in saved.items():
            setattr(Config, k, v)
