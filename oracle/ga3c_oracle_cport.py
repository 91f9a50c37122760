"""ctypes wrapper of oracle/ga3c_oracle_c.c (TEST INFRASTRUCTURE: tests/ and bench.py cpu_baseline only)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libga3c_oracle.so")
_f = C.POINTER(C.c_float)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise RuntimeError("%s missing: run `make -C oracle`" % _LIB)
        _lib = C.CDLL(_LIB)
        _lib.ga3c_oc_param_count.restype = C.c_int64
        _lib.ga3c_oc_param_count.argtypes = [C.c_int]
        _lib.ga3c_oc_max_threads.restype = C.c_int
        _lib.ga3c_oc_set_threads.argtypes = [C.c_int]
        _lib.ga3c_oc_predict.argtypes = [_f, C.c_int, _f, C.c_int, _f, _f]
        _lib.ga3c_oc_train.argtypes = [_f, _f, _f, C.c_int, _f, _f, _f, C.c_int, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_float, _f]
    return _lib


def _p(a):
    return a.ctypes.data_as(_f)


def threads():
    return lib().ga3c_oc_max_threads()


def predict(theta, num_actions, x):
    x = np.ascontiguousarray(x, np.float32)
    b = x.shape[0]
    p = np.empty((b, num_actions), np.float32)
    v = np.empty(b, np.float32)
    if lib().ga3c_oc_predict(_p(theta), num_actions, _p(x), b, _p(p), _p(v)) != 0:
        raise MemoryError
    return p, v


def train(theta, ms, num_actions, x, y_r, a, lr, beta, log_eps=1e-6, min_policy=0.0, rho=0.99, eps=0.1):
    """In-place step on theta/ms (lr < 0: gradients only).  Returns (losses[3], grad arena)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y_r, np.float32)
    a = np.ascontiguousarray(a, np.float32)
    grad = np.empty_like(theta)
    losses = np.empty(3, np.float32)
    rc = lib().ga3c_oc_train(_p(theta), _p(ms), _p(grad), num_actions, _p(x), _p(y), _p(a), x.shape[0], lr, beta,
                             log_eps, min_policy, rho, eps, _p(losses))
    if rc != 0:
        raise MemoryError
    return losses, grad
