"""Device-resident solver object over the C ABI (include/gato_hip.h).

Mirrors the reference's per-solve driver gato_linsys (gpu_library.cu:25-83) and the three
launch wrappers it calls - form_schur (src/gato_schur.cuh:885-1009), solve_pcg
(src/gato_pcg.cuh:476-567), compute_dz (src/gato_schur.cuh:1012-1022) - with the same stage
names, on device buffers.  torch is used only to hold device memory and name the stream.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np
import torch

from . import _lib

_TORCH_DT = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_cuda and t.is_contiguous()
        return ct.c_void_p(t.data_ptr())
    return ct.c_void_p(int(t))


class Solver:
    """Workspace + kernels for one (STATE_SIZE, CONTROL_SIZE, KNOT_POINTS, dtype) on one GPU."""

    def __init__(self, S: int, C: int, K: int, dtype=np.float32, device: int = 0, batch: int = 1):
        self.S, self.C, self.K = int(S), int(C), int(K)
        self.batch = int(batch)
        self.np_dtype = np.dtype(dtype)
        self.dtype = _TORCH_DT[self.np_dtype]
        self.device = int(device)
        self._h = ct.c_void_p()
        code = _lib.GATO_F32 if self.np_dtype == np.float32 else _lib.GATO_F64
        _lib.check(_lib.lib().gato_solver_create_batched(self.S, self.C, self.K, self.batch, code, self.device,
                                                         ct.byref(self._h)))
        self.n = self.S + self.C
        self.N = self.n * self.K - self.C
        self.sizes = dict(G_dense=(S * S + C * C) * K - C * C, C_dense=(S * S + S * C) * (K - 1),
                          bd=3 * S * S * K, sk=S * K)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().gato_solver_destroy(self._h)
            self._h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- options -------------------------------------------------------------------------
    def set_option(self, name: str, value: int):
        _lib.check(_lib.lib().gato_solver_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = ct.c_int()
        _lib.check(_lib.lib().gato_solver_get_option(self._h, name.encode(), ct.byref(v)))
        return v.value

    def tune(self):
        """Measure the hosting XCD of the one-XCD launches for the geometry the CURRENT options plan (blocking, ~1 ms;
        gato_solver_tune).  The constructor did it for the default geometry."""
        _lib.check(_lib.lib().gato_solver_tune(self._h, self._stream()))

    def _stream(self):
        return ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def new(self, n, dtype=None):
        return torch.empty(int(n), dtype=dtype or self.dtype, device=f"cuda:{self.device}")

    def to_device(self, a, dtype=None):
        a = np.ascontiguousarray(a, dtype or self.np_dtype)
        return torch.from_numpy(a).to(f"cuda:{self.device}")

    # ---- stages (device tensors in, device tensors out) -------------------------------------
    def convert(self, G_row, G_col, G_val, C_row, C_col, C_val, rho):
        Gd, Cd = self.new(self.sizes["G_dense"]), self.new(max(self.sizes["C_dense"], 1))
        _lib.check(_lib.lib().gato_convert(self._h, _ptr(G_row), _ptr(G_col), _ptr(G_val), _ptr(C_row),
                                           _ptr(C_col), _ptr(C_val), float(rho), _ptr(Gd), _ptr(Cd),
                                           self._stream()))
        return Gd, Cd[: self.sizes["C_dense"]]

    def form_schur(self, Gd, Cd, g, c):
        Sb, Pb = self.new(self.sizes["bd"]), self.new(self.sizes["bd"])
        gam, Gi = self.new(self.sizes["sk"]), self.new(self.sizes["G_dense"])
        _lib.check(_lib.lib().gato_form_schur(self._h, _ptr(Gd), _ptr(Cd), _ptr(g), _ptr(c), _ptr(Sb),
                                              _ptr(Pb), _ptr(gam), _ptr(Gi), self._stream()))
        return Sb, Pb, gam, Gi

    def form_ss(self, Sb, Pb):
        _lib.check(_lib.lib().gato_form_ss(self._h, _ptr(Sb), _ptr(Pb), self._stream()))
        return Pb

    def pcg(self, Sb, Pb, gamma, exit_tol, max_iters, lam=None, iters=None, check=True):
        """lam: output; with set_option("true_warm_start", 1) it is also the initial guess (in place)."""
        lam = self.new(self.sizes["sk"]) if lam is None else lam
        iters = self.new(1, torch.int32) if iters is None else iters
        _lib.check(_lib.lib().gato_pcg(self._h, _ptr(Sb), _ptr(Pb), _ptr(gamma), _ptr(lam), float(exit_tol),
                                       int(max_iters), _ptr(iters), self._stream()))
        if check:
            torch.cuda.current_stream(self.device).synchronize()
            _lib.check(_lib.lib().gato_pcg_status(self._h, None))
        return lam, iters

    def compute_dz(self, Gi, Cd, g, lam):
        dz = self.new(self.N)
        _lib.check(_lib.lib().gato_compute_dz(self._h, _ptr(Gi), _ptr(Cd), _ptr(g), _ptr(lam), _ptr(dz),
                                              self._stream()))
        return dz

    # ---- whole solve on device-resident CSR (gato_linsys, gpu_library.cu:25-83) ----------------
    def linsys(self, G_row, G_col, G_val, C_row, C_col, C_val, g, c, exit_tol, max_iters, rho,
               lam=None, dz=None):
        _lib.check(_lib.lib().gato_linsys_device(self._h, _ptr(G_row), _ptr(G_col), _ptr(G_val), _ptr(C_row),
                                                 _ptr(C_col), _ptr(C_val), _ptr(g), _ptr(c), float(exit_tol),
                                                 int(max_iters), float(rho), _ptr(lam), _ptr(dz),
                                                 self._stream()))

    def linsys_blocks(self, G_blocks, C_blocks, g, c, exit_tol, max_iters, rho, lam=None, dz=None):
        """Direct block input: G_blocks / C_blocks in the G_dense / C_dense layouts, rho not yet added."""
        _lib.check(_lib.lib().gato_linsys_device_blocks(self._h, _ptr(G_blocks), _ptr(C_blocks), _ptr(g), _ptr(c),
                                                        float(exit_tol), int(max_iters), float(rho), _ptr(lam),
                                                        _ptr(dz), self._stream()))

    def linsys_batched(self, G_row, G_col, G_val, C_row, C_col, C_val, g, c, exit_tol, max_iters, rho,
                       lam, dz, iters=None):
        """B systems with a shared CSR structure: G_val [B*nnzG], C_val [B*nnzC], g [B*N], c [B*S*K]."""
        nnzG, nnzC = G_val.numel() // self.batch, C_val.numel() // self.batch
        _lib.check(_lib.lib().gato_linsys_device_batched(
            self._h, _ptr(G_row), _ptr(G_col), _ptr(G_val), nnzG, _ptr(C_row), _ptr(C_col), _ptr(C_val), nnzC,
            _ptr(g), _ptr(c), float(exit_tol), int(max_iters), float(rho), _ptr(lam), _ptr(dz), _ptr(iters),
            self._stream()))

    def upload_batch(self, systems):
        """list of KKTSystem with identical sparsity -> device tensors in linsys_batched() argument order."""
        s0 = systems[0]
        for s in systems:
            assert np.array_equal(s.G_row, s0.G_row) and np.array_equal(s.G_col, s0.G_col)
            assert np.array_equal(s.C_row, s0.C_row) and np.array_equal(s.C_col, s0.C_col)
        dev = f"cuda:{self.device}"
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
        cat = lambda name: np.concatenate([getattr(s, name) for s in systems])
        return (t(s0.G_row, np.int32), t(s0.G_col, np.int32), t(cat("G_val"), self.np_dtype),
                t(s0.C_row, np.int32), t(s0.C_col, np.int32), t(cat("C_val"), self.np_dtype),
                t(cat("g"), self.np_dtype), t(cat("c"), self.np_dtype))

    def buffer_ptr(self, which: int) -> int:
        return int(_lib.lib().gato_solver_buffer(self._h, which))

    _BUFFERS = dict(G_dense=0, C_dense=1, Ginv=2, S=3, Pinv=4, gamma=5, lam=6, dz=7)

    def read_buffer(self, name: str):
        """Host copy of one of the solver's own work buffers (all systems of a batch), for tests and debugging."""
        which = self._BUFFERS[name]
        n = {0: self.sizes["G_dense"], 1: self.sizes["C_dense"], 2: self.sizes["G_dense"], 3: self.sizes["bd"],
             4: self.sizes["bd"], 5: self.sizes["sk"], 6: self.sizes["sk"], 7: self.N}[which] * self.batch
        out = np.empty(n, self.np_dtype)
        if n == 0:
            return out
        torch.cuda.synchronize(self.device)
        rc = ct.CDLL("libamdhip64.so").hipMemcpy(out.ctypes.data_as(ct.c_void_p), ct.c_void_p(self.buffer_ptr(which)),
                                                 ct.c_size_t(out.nbytes), 2)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")
        return out

    def pcg_last_ms(self) -> float:
        """Device time of the last PCG launch (needs set_option("time_pcg", 1))."""
        ms = ct.c_float()
        _lib.check(_lib.lib().gato_pcg_last_ms(self._h, ct.byref(ms)))
        return ms.value

    def last_stage_ms(self):
        """{assembly, pcg, dz} device times (ms) of the last linsys / linsys_blocks call (needs set_option("time_stages", 1))."""
        ms = (ct.c_float * 3)()
        _lib.check(_lib.lib().gato_last_stage_ms(self._h, ms))
        return dict(assembly=ms[0], pcg=ms[1], dz=ms[2])

    def eta_history(self, n: int):
        """eta = r . Pinv r after the initial step and after each of the first n iterations (needs record_eta=1)."""
        ptr = int(_lib.lib().gato_solver_buffer(self._h, 10))
        buf = (ct.c_double * (n + 1))()
        hip = ct.CDLL("libamdhip64.so")
        torch.cuda.synchronize(self.device)
        rc = hip.hipMemcpy(buf, ct.c_void_p(ptr), 8 * (n + 1), 2)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")
        return np.frombuffer(buf, dtype=np.float64).copy()

    def check_status(self):
        """Raises GatoError(ETIMEOUT) if a hand-off of any PCG launch since the last check timed out."""
        _lib.check(_lib.lib().gato_pcg_status(self._h, None))

    def recover(self) -> bool:
        """After linsys / linsys_blocks: if a persistent launch timed out (its workgroups were not co-resident), re-run
        the PCG through the streaming kernels and recompute dz into the same buffers.  True if that happened."""
        v = ct.c_int()
        _lib.check(_lib.lib().gato_solver_recover(self._h, ct.byref(v), self._stream()))
        return bool(v.value)

    def upload_system(self, sysm):
        """KKTSystem (host CSR) -> tuple of device tensors in linsys() argument order."""
        i32 = torch.int32
        dev = f"cuda:{self.device}"
        t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
        return (t(sysm.G_row, np.int32), t(sysm.G_col, np.int32), t(sysm.G_val, self.np_dtype),
                t(sysm.C_row, np.int32), t(sysm.C_col, np.int32), t(sysm.C_val, self.np_dtype),
                t(sysm.g, self.np_dtype), t(sysm.c, self.np_dtype))
