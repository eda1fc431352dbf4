"""ORACLE - test infrastructure, not product code.  ctypes view of oracle/libgato_oracle.so
(the C restatement, oracle/gato_oracle_impl.h).  Same call shapes as oracle/gato_oracle.py."""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgato_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ct.CDLL(build())
        _LIB.gato_oracle_pcg_f32.restype = ct.c_int
        _LIB.gato_oracle_pcg_f64.restype = ct.c_int
        _LIB.gato_oracle_linsys_f32.restype = ct.c_int
        _LIB.gato_oracle_linsys_f64.restype = ct.c_int
        _LIB.gato_oracle_max_threads.restype = ct.c_int
    return _LIB


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", ct.c_float
    if dtype == np.float64:
        return "f64", ct.c_double
    raise TypeError(dtype)


def _p(a):
    return a.ctypes.data_as(ct.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def set_threads(n: int):
    lib().gato_oracle_set_threads(int(n))


def max_threads() -> int:
    return lib().gato_oracle_max_threads()


def convert(G_row, G_col, G_val, C_row, C_col, C_val, S, C, K, rho, dtype=np.float64):
    suf, cT = _suf(dtype)
    Gd = np.zeros((S * S + C * C) * K - C * C, dtype)
    Cd = np.zeros((S * S + S * C) * (K - 1), dtype)
    a = [_i32(G_row), _i32(G_col), np.ascontiguousarray(G_val, dtype),
         _i32(C_row), _i32(C_col), np.ascontiguousarray(C_val, dtype)]
    getattr(lib(), "gato_oracle_convert_" + suf)(*[_p(x) for x in a], S, C, K, cT(rho), _p(Gd), _p(Cd))
    return Gd, Cd


def form_schur(Gd, Cd, g, c, S, C, K):
    dtype = Gd.dtype
    suf, _ = _suf(dtype)
    Sbd = np.zeros(3 * S * S * K, dtype)
    Pbd = np.zeros(3 * S * S * K, dtype)
    gam = np.zeros(S * K, dtype)
    Gi = np.zeros_like(Gd)
    g = np.ascontiguousarray(g, dtype)
    c = np.ascontiguousarray(c, dtype)
    getattr(lib(), "gato_oracle_form_schur_" + suf)(_p(Gd), _p(Cd), _p(g), _p(c), S, C, K,
                                                     _p(Sbd), _p(Pbd), _p(gam), _p(Gi))
    return Sbd, Pbd, gam, Gi


def form_ss(Sbd, Pbd, S, K):
    suf, _ = _suf(Sbd.dtype)
    P = Pbd.copy()
    getattr(lib(), "gato_oracle_form_ss_" + suf)(_p(Sbd), _p(P), S, K)
    return P


def pcg(Sbd, Pbd, gamma, S, K, exit_tol, max_iters, return_history=False):
    dtype = Sbd.dtype
    suf, cT = _suf(dtype)
    lam = np.zeros(S * K, dtype)
    hist = np.full(max_iters + 1, np.nan, dtype)
    gamma = np.ascontiguousarray(gamma, dtype)
    it = getattr(lib(), "gato_oracle_pcg_" + suf)(_p(Sbd), _p(Pbd), _p(gamma), S, K, cT(exit_tol),
                                                   int(max_iters), _p(lam), _p(hist))
    if return_history:
        return lam, it, hist
    return lam, it


def compute_dz(Gi, Cd, g, lam, S, C, K):
    dtype = Gi.dtype
    suf, _ = _suf(dtype)
    dz = np.zeros((S + C) * K - C, dtype)
    g = np.ascontiguousarray(g, dtype)
    lam = np.ascontiguousarray(lam, dtype)
    getattr(lib(), "gato_oracle_compute_dz_" + suf)(_p(Gi), _p(Cd), _p(g), _p(lam), S, C, K, _p(dz))
    return dz


def linsys_solve(G_row, G_col, G_val, C_row, C_col, C_val, g, c, S, C, K,
                 exit_tol, max_iters, rho, dtype=np.float32):
    suf, cT = _suf(dtype)
    lam = np.zeros(S * K, dtype)
    dz = np.zeros((S + C) * K - C, dtype)
    a = [_i32(G_row), _i32(G_col), np.ascontiguousarray(G_val, dtype),
         _i32(C_row), _i32(C_col), np.ascontiguousarray(C_val, dtype),
         np.ascontiguousarray(g, dtype), np.ascontiguousarray(c, dtype)]
    it = getattr(lib(), "gato_oracle_linsys_" + suf)(*[_p(x) for x in a], S, C, K, cT(exit_tol),
                                                      int(max_iters), cT(rho), _p(lam), _p(dz))
    return lam, dz, it
