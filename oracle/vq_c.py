"""ctypes front-end of ``oracle/vq_oracle.c`` -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libvq_oracle.so')
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'vq_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.vq_oracle_assign.restype = None
        _lib.vq_oracle_assign.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def assign(z: np.ndarray, e: np.ndarray, assoc: int = 0):
    """z [N,D], e [K,D] float32 -> (idx int64 [N], dmin f32 [N], z2 [N], e2 [K]) in the canonical order."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    e = np.ascontiguousarray(e, dtype=np.float32)
    n, d = z.shape
    k = e.shape[0]
    assert e.shape[1] == d and d % 8 == 0
    idx = np.empty(n, np.int64)
    dmin = np.empty(n, np.float32)
    z2 = np.empty(n, np.float32)
    e2 = np.empty(k, np.float32)
    _load().vq_oracle_assign(z.ctypes.data, e.ctypes.data, n, k, d, assoc, idx.ctypes.data, dmin.ctypes.data,
                             z2.ctypes.data, e2.ctypes.data)
    return idx, dmin, z2, e2
