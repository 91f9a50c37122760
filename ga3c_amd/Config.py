"""Global settings of the GA3C engine, read as class attributes exactly like the reference's
Config (/root/reference/ga3c/Config.py:27-202) and overridable from argv as KEY=VALUE
(GA3C.py:39-43).  Names and meanings are the reference's; defaults are the ones SURVEY.md §8-d
fixes for the 84x84x4 Atari path (the fork's own defaults target Pendulum), and the block at the
end adds the knobs this engine needs (devices, transport, return mode).
"""


class Config:
    # ---- what to run -------------------------------------------------------------------------
    GAME = 'PongDeterministic-v4'       # gym id, used by FRAME_SOURCE = 'gym' only; the offline sources are synthetic
    PLAY_MODE = False                   # greedy actions, no training, one agent (GA3C.py:46-54)
    TRAIN_MODELS = True
    LOAD_CHECKPOINT = False
    LOAD_EPISODE = 0                    # 0 = latest checkpoint

    # ---- workers (initial values when DYNAMIC_SETTINGS is on) ---------------------------------
    AGENTS = 32
    HUMAN_REF_AGENTS = 0                # pyperrace-only in the reference; kept for argv compatibility
    PREDICTORS = 2
    TRAINERS = 2
    DEVICE = 'gpu:0'
    DYNAMIC_SETTINGS = False            # ThreadDynamicAdjustment random walk over NT/NP/NA
    DYNAMIC_SETTINGS_STEP_WAIT = 20
    DYNAMIC_SETTINGS_INITIAL_WAIT = 10

    # ---- algorithm ----------------------------------------------------------------------------
    DISCOUNTING = True
    DISCOUNT = 0.99
    TIME_MAX = 5                        # rollout cut (upstream Atari value, Config.py:77 comment)
    REWARD_CLIPPING = True
    USE_INTERMEDIATE_REWARD = False
    REWARD_MIN = -1
    REWARD_MAX = 1
    MAX_QUEUE_SIZE = 100
    PREDICTION_BATCH_SIZE = 128
    STACKED_FRAMES = 4
    IMAGE_WIDTH = 84
    IMAGE_HEIGHT = 84
    EPISODES = 400000
    ANNEALING_EPISODE_COUNT = 400000
    BETA_START = 0.01
    BETA_END = 0.01
    LEARNING_RATE_START = 0.0003
    LEARNING_RATE_END = 0.0003
    RMSPROP_DECAY = 0.99
    RMSPROP_MOMENTUM = 0.0
    RMSPROP_EPSILON = 0.1
    DUAL_RMSPROP = False                # out of scope; must stay False
    USE_GRAD_CLIP = False
    GRAD_CLIP_NORM = 40.0
    LOG_EPSILON = 1e-6
    TRAINING_MIN_BATCH_SIZE = 0
    MIN_POLICY = 0.0
    USE_LOG_SOFTMAX = False
    DISCRATE_INPUT = True               # (sic) discrete action space: softmax policy head
    CONTINUOUS_INPUT = False
    USE_DDPG = False
    USE_REPLAY_MEMORY = False
    USE_NETWORK_TESTER = False
    RANDOM_SEED = 12345

    # ---- logging / checkpoints ----------------------------------------------------------------
    TENSORBOARD = False                 # scalar log written as CSV under logs/<NETWORK_NAME>/
    TENSORBOARD_UPDATE_FREQUENCY = 1000
    SAVE_MODELS = True
    SAVE_FREQUENCY = 1000
    PRINT_STATS_FREQUENCY = 1
    STAT_ROLLING_MEAN_WINDOW = 1000
    RESULTS_FILENAME = 'results.txt'
    NETWORK_NAME = 'network'

    # ---- engine knobs (no counterpart in the reference) ----------------------------------------
    NUM_ACTIONS = 6                     # synthetic source only (Pong 6, Breakout 4, Boxing 18)
    MAX_SECONDS = 0                     # > 0: Server.main stops after this many seconds (the reference stops on EPISODES only)
    RETURN_MODE = 'fork'                # 'fork': ProcessAgent.py:69-84 bit-exact; 'nstep': upstream n-step
    STATE_TRANSPORT = 'u8'              # 'u8': ship uint8 frames, convert on GPU; 'f32': ship f32 states
    SYNTHETIC_EPISODE_LENGTH = 1000
    TRAIN_ROWS_MAX = 0                  # capacity of one train call; 0 = derive from the batch knobs
    HOGWILD = False                     # True: TRAINERS train lanes update the weights concurrently and unlocked,
                                        # as the reference's trainer threads do; False: synchronous steps (default)
    ZERO_COPY = True                    # GPU gathers states straight from the registered shm transport
    QUEUE_TIMEOUT_MS = 200              # workers re-check their exit flag this often
    NATIVE_PREDICTOR = True             # ThreadPredictor's loop in native code (ga3c_pq_serve) when ZERO_COPY is on
    PIPELINED_PREDICTOR = True          # ... answering batch k beside the GPU's work on batch k+1 (ga3c_pq_serve_pipelined)
    PIPELINED_FRAMES = False            # the same overlap for the frames loop (device frame queue): pays from ~500 agents per GPU on
    STATE_CACHE = True                  # the engine keeps the uint8 states its predictions read (a ring per agent in HBM) and
                                        # rollouts NAME their states (agent, request number) instead of carrying them: no second
                                        # trip over PCIe for training (needs ZERO_COPY, STATE_TRANSPORT = 'u8', the native
                                        # pipelined predictor and trainer loops; Server falls back without them)
    STATE_CACHE_DEPTH = 0               # states kept per agent (28,224 B each); 0 = four times an agent's fair share of the rows
                                        # in flight plus four rollouts, at least 64 (2.3 MB per agent at the defaults)
    STATE_CACHE_ACTIVE = False          # (set by Server for its agents: the cache is really in use)
    NATIVE_TRAINER = True               # ThreadTrainer's batch assembly in one native call (ga3c_tq_collect) when ZERO_COPY is on
    CPU_AFFINITY = 'auto'               # where server threads and agents run (Placement.py): 'auto' = as many CPUs as the cgroup's
                                        # quota allows, whole L3 domains next to the GPU; 'off'; or a list such as '0-15'
    AGENT_CPUS = None                   # (set by Server when the placement keeps CPUs apart for the agents)
    AGENT_SPIN_US = 0                   # > 0: an agent polls this long for its answer before it sleeps on the slot's futex
    PREDICTION_LINGER_US = 0            # > 0: a predictor holding fewer than PREDICTION_LINGER_BATCH requests after its
    PREDICTION_LINGER_BATCH = 0         # greedy drain keeps collecting this long (the reference never waits: 0)
    ROLLOUT_SLOTS = 0                   # rollout slots of the transport; 0 = MAX_QUEUE_SIZE (the reference's queue bound)
                                        # plus what the trainers keep while a zero-copy batch fills and trains
    FRAME_SOURCE = 'planes'             # 'gym': gym.make(GAME) frames (needs gym + ALE, absent offline: untested here);
                                        # 'planes': synthetic 84x84 uint8 planes; 'rgb': synthetic
                                        # emulator frames FRAME_HEIGHT x FRAME_WIDTH x 3 that go through the reference's
                                        # front-end (Environment.py:52-74: gray, bytescale, bilinear resize, frame queue)
    FRONTEND = 'host'                   # where that front-end runs for 'rgb' frames: 'host' = in the agent process
                                        # (ga3c_frame_preprocess), states shipped as before; 'device' = the agent ships
                                        # the raw frame, planes / frame queues / training rows stay in HBM.  With 'planes'
                                        # 'device' keeps only the 4-deep frame queue and the plane history in HBM: the agent
                                        # ships its newest 84x84 plane (7,056 B per step instead of a 28,224 B state, and
                                        # nothing at all for training)
    FRAME_HEIGHT = 210
    FRAME_WIDTH = 160
    FRAME_HISTORY = 0                   # planes of history per agent on the device; 0 = derived from the queue bounds
