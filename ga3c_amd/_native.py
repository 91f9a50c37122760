"""ctypes bindings of the two product libraries (include/ga3c_abi.h, include/ga3c_host.h).

There is no fallback: if a library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "libga3c_hip.so")
HOST_LIB = os.path.join(_HERE, "libga3c_host.so")

STATE_FLOATS = 84 * 84 * 4
COMM_ID_BYTES = 128
FLAG_LOG_SOFTMAX = 1
FLAG_GRAD_CLIP = 2

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


class NetConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("num_actions", C.c_int32), ("max_batch", C.c_int32), ("flags", C.c_uint32),
                ("rmsprop_decay", C.c_float), ("rmsprop_momentum", C.c_float), ("rmsprop_epsilon", C.c_float),
                ("log_epsilon", C.c_float), ("min_policy", C.c_float), ("grad_clip_norm", C.c_float),
                ("predict_lanes", C.c_int32), ("train_lanes", C.c_int32)]


class ShmConfig(C.Structure):
    _fields_ = [("max_agents", C.c_int32), ("num_actions", C.c_int32), ("state_bytes", C.c_int32),
                ("train_slots", C.c_int32), ("train_rows", C.c_int32), ("rollout_row_bytes", C.c_int32),
                ("reserved", C.c_int32 * 2)]


class ServeStats(C.Structure):     # include/ga3c_host.h: ga3c_serve_stats
    _fields_ = [("batches", C.c_int64), ("served", C.c_int64), ("ns_pop", C.c_int64), ("ns_predict", C.c_int64),
                ("ns_respond", C.c_int64), ("largest_batch", C.c_int64), ("reserved", C.c_int64 * 2)]


HIP_SIGNATURES = {
    "ga3c_last_error": (C.c_char_p, []),
    "ga3c_device_count": (C.c_int, [i32p]),
    "ga3c_device_pci_bus_id": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "ga3c_net_create": (C.c_int, [C.POINTER(NetConfig), C.POINTER(C.c_void_p)]),
    "ga3c_net_destroy": (C.c_int, [C.c_void_p]),
    "ga3c_net_param_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ga3c_net_get_arena": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_int64]),
    "ga3c_net_set_arena": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_int64]),
    "ga3c_net_num_params": (C.c_int32, [C.c_void_p]),
    "ga3c_net_param_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "ga3c_net_param_info": (C.c_int, [C.c_void_p, C.c_char_p, i64p, i64p, i32p, i64p]),
    "ga3c_net_get_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, f32p, C.c_int64]),
    "ga3c_net_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, f32p, C.c_int64]),
    "ga3c_net_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ga3c_net_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "ga3c_net_get_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "ga3c_net_set_step": (C.c_int, [C.c_void_p, C.c_int64]),
    "ga3c_net_predict": (C.c_int, [C.c_void_p, f32p, C.c_int32, f32p, f32p, f32p]),
    "ga3c_net_predict_u8": (C.c_int, [C.c_void_p, u8p, C.c_int32, f32p, f32p, f32p]),
    "ga3c_net_train": (C.c_int, [C.c_void_p, f32p, f32p, f32p, C.c_int32, C.c_float, C.c_float, f32p]),
    "ga3c_net_train_u8": (C.c_int, [C.c_void_p, u8p, f32p, f32p, C.c_int32, C.c_float, C.c_float, f32p]),
    "ga3c_net_evaluate": (C.c_int, [C.c_void_p, f32p, u8p, i64p, C.c_int32, f32p, f32p, C.c_int32, C.c_float, f32p, f32p, f32p,
                                    f32p]),
    "ga3c_net_evaluate_frames": (C.c_int, [C.c_void_p, i32p, i64p, f32p, f32p, C.c_int32, C.c_float, f32p, f32p, f32p, f32p]),
    "ga3c_net_compute_grads": (C.c_int, [C.c_void_p, f32p, f32p, f32p, C.c_int32, C.c_float, f32p]),
    "ga3c_net_apply_grads": (C.c_int, [C.c_void_p, C.c_float]),
    "ga3c_net_register_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "ga3c_net_unregister_host": (C.c_int, [C.c_void_p]),
    "ga3c_net_predict_gather": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, f32p, f32p, f32p]),
    "ga3c_net_predict_gather_begin": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, i32p]),
    "ga3c_net_predict_gather_end": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, f32p, f32p]),
    "ga3c_net_train_gather": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, f32p, f32p, C.c_int32, C.c_float,
                                        C.c_float, f32p]),
    "ga3c_net_frames_config": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "ga3c_net_frames_preprocess": (C.c_int, [C.c_void_p, u8p, C.c_int32, u8p]),
    "ga3c_net_frames_push": (C.c_int, [C.c_void_p, u8p, i32p, u8p, C.c_int32, i64p]),
    "ga3c_net_frames_push_offsets": (C.c_int, [C.c_void_p, i64p, i32p, u8p, C.c_int32, i64p]),
    "ga3c_net_serve_frames": (C.c_int, [C.c_void_p, i64p, i32p, u32p, C.c_int32, f32p, f32p]),
    "ga3c_net_serve_frames_begin": (C.c_int, [C.c_void_p, i64p, i32p, u32p, C.c_int32, i32p]),
    "ga3c_net_serve_frames_end": (C.c_int, [C.c_void_p, C.c_int32, u32p, C.c_int32, f32p, f32p]),
    "ga3c_net_train_frames": (C.c_int, [C.c_void_p, i32p, i64p, f32p, f32p, C.c_int32, C.c_float, C.c_float, f32p]),
    "ga3c_net_train_cached": (C.c_int, [C.c_void_p, i32p, i64p, f32p, f32p, C.c_int32, C.c_float, C.c_float, f32p]),
    "ga3c_net_evaluate_cached": (C.c_int, [C.c_void_p, i32p, i64p, f32p, f32p, C.c_int32, C.c_float, f32p, f32p, f32p, f32p]),
    "ga3c_net_state_cache_config": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "ga3c_net_predict_gather_begin_cached": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), i32p, i64p, C.c_int32, C.c_int32, i32p]),
    "ga3c_net_frames_state": (C.c_int, [C.c_void_p, C.c_int32, u8p, i32p]),
    "ga3c_net_predict_frames": (C.c_int, [C.c_void_p, i32p, C.c_int32, f32p, f32p, f32p]),
    "ga3c_net_frames_pushed": (C.c_int, [C.c_void_p, C.c_int32, i64p]),
    "ga3c_net_frames_upload": (C.c_int, [C.c_void_p, u8p, C.c_int32]),
    "ga3c_net_time_frames": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, f32p]),
    "ga3c_net_upload": (C.c_int, [C.c_void_p, f32p, f32p, f32p, C.c_int32]),
    "ga3c_net_upload_u8": (C.c_int, [C.c_void_p, u8p, f32p, f32p, C.c_int32]),
    "ga3c_net_predict_resident": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_net_train_resident": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float]),
    "ga3c_net_sync": (C.c_int, [C.c_void_p]),
    "ga3c_net_time_resident": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, f32p]),
    "ga3c_net_time_predict_lanes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, f32p]),
    "ga3c_net_time_train_lanes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, f32p]),
    "ga3c_net_time_kernel": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, f32p]),
    "ga3c_net_fetch": (C.c_int, [C.c_void_p, C.c_char_p, f32p, C.c_int64]),
    "ga3c_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64]),
    "ga3c_host_free": (C.c_int, [C.c_void_p]),
    "ga3c_comm_make_id": (C.c_int, [u8p]),
    "ga3c_net_comm_init": (C.c_int, [C.c_void_p, u8p, C.c_int32, C.c_int32]),
    "ga3c_net_last_lanes_gpu_ms": (C.c_int, [C.c_void_p, f32p]),
    "ga3c_net_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32]),
    "ga3c_net_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ga3c_net_allreduce_grads": (C.c_int, [C.c_void_p]),
    "ga3c_net_time_allreduce": (C.c_int, [C.c_void_p, C.c_int32, f32p]),
}

HOST_SIGNATURES = {
    "ga3c_host_last_error": (C.c_char_p, []),
    "ga3c_returns_fork": (C.c_int, [f64p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32, f64p]),
    "ga3c_returns_nstep": (C.c_int, [f64p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, f64p]),
    "ga3c_shm_create": (C.c_int, [C.c_char_p, C.POINTER(ShmConfig), C.POINTER(C.c_void_p)]),
    "ga3c_shm_attach": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "ga3c_shm_close": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_shm_unlink": (C.c_int, [C.c_void_p]),
    "ga3c_shm_shutdown": (C.c_int, [C.c_void_p]),
    "ga3c_shm_base": (C.c_void_p, [C.c_void_p]),
    "ga3c_shm_bytes": (C.c_int64, [C.c_void_p]),
    "ga3c_shm_get_config": (C.c_int, [C.c_void_p, C.POINTER(ShmConfig)]),
    "ga3c_shm_state_offset": (C.c_int64, [C.c_void_p, C.c_int32]),
    "ga3c_shm_agent_stride": (C.c_int64, [C.c_void_p]),
    "ga3c_shm_rollout_offset": (C.c_int64, [C.c_void_p, C.c_int32]),
    "ga3c_shm_rollout_stride": (C.c_int64, [C.c_void_p]),
    "ga3c_pq_state_ptr": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "ga3c_pq_submit": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_pq_submit_flags": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint32]),
    "ga3c_pq_request_flags": (C.c_int, [C.c_void_p, u32p, C.c_int32, u32p]),
    "ga3c_pq_wait": (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, C.c_int32]),
    "ga3c_select_action": (C.c_int32, [f32p, C.c_int32, C.c_double]),
    "ga3c_pq_serve_pipelined": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ga3c_pq_serve_pipelined_cached": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ga3c_pq_request_seq": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "ga3c_frame_queue_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),   # (plane: address or bytes)
    "ga3c_pq_agent_idle": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_pq_pop_batch": (C.c_int, [C.c_void_p, u32p, C.c_int32, C.c_int32]),
    "ga3c_pq_respond": (C.c_int, [C.c_void_p, u32p, C.c_int32, f32p, f32p]),
    "ga3c_frame_preprocess": (C.c_int, [u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, u8p]),
    "ga3c_pq_set_linger": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "ga3c_pq_set_spin": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_host_signal_hold": (C.c_int, [C.c_int32]),
    "ga3c_pq_wake_latency": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ga3c_pq_round_trip": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_uint32, C.c_int32, C.c_int32, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "ga3c_pq_serve_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ga3c_pq_serve_frames_pipelined": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ga3c_pq_serve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ga3c_tq_acquire": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_tq_states": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "ga3c_tq_returns": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "ga3c_tq_actions": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "ga3c_tq_commit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "ga3c_tq_pop": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_tq_rows": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_tq_release": (C.c_int, [C.c_void_p, C.c_int32]),
    "ga3c_tq_collect": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ga3c_tq_release_many": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "ga3c_tq_ready_count": (C.c_int, [C.c_void_p]),
    "ga3c_tq_free_count": (C.c_int, [C.c_void_p]),
}

_libs = {}


def _load(path, sigs):
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: build it with `make -C ga3c_amd/csrc` "
                           "(or python -c 'import __graft_entry__ as g; g.build()')" % path)
    lib = C.CDLL(path)
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if the ABI and the header disagree
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib


def hip_lib():
    return _load(HIP_LIB, HIP_SIGNATURES)


def host_lib():
    return _load(HOST_LIB, HOST_SIGNATURES)


ELOST = -5         # GA3C_ELOST: a named row is no longer in the state cache


class StateLost(RuntimeError):
    """A train / evaluate batch named a state the engine no longer holds (GA3C_ELOST): nothing was trained."""


def check(rc, what="ga3c call"):
    if rc == ELOST:
        raise StateLost("%s: %s" % (what, (hip_lib().ga3c_last_error() or b"").decode()))
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, (hip_lib().ga3c_last_error() or b"").decode()))
    return rc


def check_host(rc, what="ga3c host call"):
    if rc < 0 and rc not in (-3, -4):
        raise RuntimeError("%s failed (%d): %s" % (what, rc, (host_lib().ga3c_host_last_error() or b"").decode()))
    return rc


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a, typ=f32p):
    return a.ctypes.data_as(typ)


def pinned_array(shape, dtype=np.float32):
    """numpy array over hipHostMalloc memory (DMA-able staging, cf. ThreadPredictor.py:46-47)."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    check(hip_lib().ga3c_host_alloc(C.byref(p), nbytes), "ga3c_host_alloc")
    buf = (C.c_uint8 * nbytes).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    return arr, p


def free_pinned(p):
    check(hip_lib().ga3c_host_free(p), "ga3c_host_free")
