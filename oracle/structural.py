"""numpy front-end of oracle/structural_oracle.c (ctypes).  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "structural_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def set_threads(n):
    """host threads of the Chamfer restatement (independent queries; the results do not depend on it).  Default 1."""
    lib().oracle_set_threads(int(n))


def nndistance(xyz1, xyz2):
    """(B,n,3),(B,m,3) -> dist1 (B,n), idx1 (B,n) int32, dist2 (B,m), idx2 (B,m)."""
    xyz1, xyz2 = _c(xyz1), _c(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((b, n), np.float32); i1 = np.empty((b, n), np.int32)
    d2 = np.empty((b, m), np.float32); i2 = np.empty((b, m), np.int32)
    lib().oracle_nndistance(b, n, _fp(xyz1), m, _fp(xyz2), _fp(d1), _ip(i1), _fp(d2), _ip(i2))
    return d1, i1, d2, i2


def nndistancegrad(xyz1, xyz2, idx1, idx2, gd1, gd2):
    xyz1, xyz2, gd1, gd2 = _c(xyz1), _c(xyz2), _c(gd1), _c(gd2)
    idx1, idx2 = _c(idx1, np.int32), _c(idx2, np.int32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((b, n, 3), np.float32); g2 = np.empty((b, m, 3), np.float32)
    lib().oracle_nndistancegrad(b, n, _fp(xyz1), m, _fp(xyz2), _fp(gd1), _ip(idx1), _fp(gd2), _ip(idx2),
                                _fp(g1), _fp(g2))
    return g1, g2


def approxmatch(xyz1, xyz2):
    """-> match (B,m,n), temp (B,2(n+m))."""
    xyz1, xyz2 = _c(xyz1), _c(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.empty((b, m, n), np.float32); temp = np.empty((b, 2 * (n + m)), np.float32)
    lib().oracle_approxmatch(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match), _fp(temp))
    return match, temp


def matchcost(xyz1, xyz2, match):
    xyz1, xyz2, match = _c(xyz1), _c(xyz2), _c(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    out = np.empty((b,), np.float32)
    lib().oracle_matchcost(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match), _fp(out))
    return out


def matchcostgrad(xyz1, xyz2, match):
    xyz1, xyz2, match = _c(xyz1), _c(xyz2), _c(match)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.empty((b, n, 3), np.float32); g2 = np.empty((b, m, 3), np.float32)
    lib().oracle_matchcostgrad(b, n, m, _fp(xyz1), _fp(xyz2), _fp(match), _fp(g1), _fp(g2))
    return g1, g2
