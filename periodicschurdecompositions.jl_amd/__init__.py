"""periodicschurdecompositions.jl_amd — MI355X-native periodic Schur engine, host-side mirror.

The reference (RalphAS/PeriodicSchurDecompositions.jl) exposes `pschur`, `pschur!`, `phessenberg!`
and the `PeriodicSchur` result type as Julia methods (src/PeriodicSchurDecompositions.jl:11,59-152).
No Julia toolchain exists in the build/GPU images, so this module mirrors that interface — same
names (`!` spelled `_`), argument meaning and error behaviour — in Python over the C ABI of
`libpsd_mi355x.so` (include/psd_mi355x.h).  The Julia `ccall` wrapper a maintainer would add is in
INTEGRATION.md.

All numerical work happens in the HIP library.  There is no CPU fallback: if the library is
missing or no GPU is visible, `Engine()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpsd_mi355x.so")
# the diagnostic build (-DPSD_DIAG: timing experiments, tuning knobs, fault injection behind environment variables).
# Only tools/ and the fault-injection test pass it as `libpath`; the package itself never loads it.
DIAG_LIB_PATH = os.path.join(_HERE, "libpsd_mi355x_diag.so")

INFO_NOCONV = 1000000
INFO_NOTIMPL = 2000000
INFO_RUNTIME = 3000000


class NotImplementedPSD(Exception):
    """PeriodicSchurDecompositions.NotImplemented (src/PeriodicSchurDecompositions.jl:30)."""


class ConvergenceError(Exception):
    """ErrorException("convergence failed at level i") (src/PeriodicSchurDecompositions.jl:892)."""

    def __init__(self, level):
        super().__init__(f"convergence failed at level {level}")
        self.level = level


class IllConditionedException(Exception):
    """IllConditionedException(info) (src/PeriodicSchurDecompositions.jl:26-28)."""

    def __init__(self, info):
        super().__init__(f"IllConditionedException({info})")
        self.info = info


class SingularException(Exception):
    """LinearAlgebra.SingularException (src/utils.jl:123-131)."""


class DimensionMismatch(ValueError):
    """DimensionMismatch (src/PeriodicSchurDecompositions.jl:216-222)."""


class Stats(C.Structure):
    _fields_ = [
        ("niter", C.c_int64), ("maxits", C.c_int32), ("nsweeps", C.c_int32), ("nrqpass", C.c_int32),
        ("ndefl1", C.c_int32), ("ndefl2", C.c_int32), ("nwindows", C.c_int32), ("nlaunch_step", C.c_int32),
        ("window", C.c_int32), ("nlog", C.c_int32), ("ms_hess", C.c_double), ("ms_formq", C.c_double),
        ("ms_iter", C.c_double), ("ms_total", C.c_double), ("ms_copy", C.c_double), ("bytes_sweeps", C.c_double),
        ("bytes_hess", C.c_double), ("bytes_formq", C.c_double), ("step_kernel_ms_avg", C.c_double),
        ("step_kernel_samples", C.c_int32), ("reserved", C.c_int32), ("step_cycles", C.c_int64 * 6),
    ]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["step_cycles"] = list(self.step_cycles)
        return d


class PeriodicSchur:
    """Mirror of the reference's `PeriodicSchur` (src/PeriodicSchurDecompositions.jl:59-92).

    `T1` is the quasi-triangular factor (position `schurindex` in the user's sequence), `T` the
    remaining p-1 upper-triangular factors in order, `Z` the p orthogonal factors (empty if
    wantZ=False), `values` the eigenvalues of the product, `orientation` 'R' or 'L'.
    """

    def __init__(self, Ts, Z, values, orientation, schurindex, stats=None, sweeplog=None):
        self.Ts = Ts  # all p factors in user order (T1 included)
        self.Z = Z
        self.values = values
        self.orientation = orientation
        self.schurindex = schurindex
        self.stats = stats
        self.sweeplog = sweeplog

    @property
    def T1(self):
        return self.Ts[self.schurindex - 1]

    @property
    def T(self):
        return [t for j, t in enumerate(self.Ts) if j != self.schurindex - 1]

    @property
    def period(self):  # src/PeriodicSchurDecompositions.jl:85-91
        return len(self.Ts)


class GeneralizedPeriodicSchur(PeriodicSchur):
    """Mirror of `GeneralizedPeriodicSchur` (src/generalized.jl:31-85): eigenvalues in scaled form
    `values = alpha ./ beta .* 2 .^ alphascale`."""

    def __init__(self, S, Ts, Z, alpha, beta, alphascale, orientation, schurindex, stats=None, sweeplog=None):
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            values = alpha / beta * np.exp2(alphascale.astype(np.float64))
        super().__init__(Ts, Z, values, orientation, schurindex, stats, sweeplog)
        self.S = list(S)
        self.alpha, self.beta, self.alphascale = alpha, beta, alphascale


def char_lr(lr):
    """src/PeriodicSchurDecompositions.jl:155-163,175-177."""
    if lr in ("R", ":R"):
        return "R"
    if lr in ("L", ":L"):
        return "L"
    raise ValueError("orientation argument must be either :R (right) or :L (left)")


def _check_square(A):
    """src/PeriodicSchurDecompositions.jl:214-222."""
    if len(A) == 0:
        raise DimensionMismatch("empty sequence")
    n = A[0].shape[0]
    for a in A:
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise DimensionMismatch("matrices must be square")
        if a.shape[0] != n:
            raise DimensionMismatch("matrices must have equal order")
    return n


class Engine:
    """One device context (stream + workspace). One call at a time per Engine."""

    def __init__(self, device=0, libpath=None):
        path = libpath or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        self.lib = lib = C.CDLL(path)
        dp, dpp = C.POINTER(C.c_double), C.POINTER(C.c_void_p)
        ip = C.POINTER(C.c_int)
        lib.psd_version.restype = C.c_char_p
        lib.psd_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        lib.psd_destroy.argtypes = [C.c_void_p]
        lib.psd_set_profile.argtypes = [C.c_void_p, C.c_int]
        lib.psd_d_phessenberg.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dp, C.POINTER(Stats), ip]
        lib.psd_d_pschur.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, C.POINTER(C.c_uint8), C.c_char, C.c_int,
                                     C.c_int, C.c_int, dpp, dp, dp, ip, C.POINTER(Stats), C.POINTER(C.c_int32),
                                     C.c_int64, ip]
        lib.psd_d_pschur_hess.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dpp, C.c_int, C.c_int, C.c_int, dp, dp,
                                          C.POINTER(Stats), C.POINTER(C.c_int32), C.c_int64, ip]
        lib.psd_d_pschur_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, dp, dp, ip, C.POINTER(Stats), C.POINTER(C.c_int32), C.c_int64, ip]
        i32p = C.POINTER(C.c_int32)
        lib.psd_z_phessenberg.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dp, C.POINTER(Stats), ip]
        lib.psd_z_pschur.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, C.POINTER(C.c_uint8), C.c_char, C.c_int,
                                     C.c_int, C.c_int, dpp, dp, dp, i32p, ip, C.POINTER(Stats), i32p, C.c_int64, ip]
        lib.psd_z_pschur_hess.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, C.POINTER(C.c_uint8), dpp, C.c_int,
                                          C.c_int, C.c_int, dp, dp, i32p, C.POINTER(Stats), i32p, C.c_int64, ip]
        lib.psd_d_gpschur_hess.argtypes = lib.psd_z_pschur_hess.argtypes
        lib.psd_d_gphessenberg.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, C.POINTER(C.c_uint8), dpp,
                                           C.POINTER(Stats), ip]
        lib.psd_z_gphessenberg.argtypes = lib.psd_d_gphessenberg.argtypes
        lib.psd_d_rphessenberg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, dpp, dpp, C.c_int, C.c_int, ip]
        lib.psd_z_rphessenberg.argtypes = lib.psd_d_rphessenberg.argtypes
        lib.psd_d_gpschur.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, C.POINTER(C.c_uint8), C.c_char, C.c_int,
                                      C.c_int, C.c_int, dpp, dp, dp, i32p, ip, C.POINTER(Stats), ip]
        lib.psd_z_pschur_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_char, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, dp, dp, i32p, ip, C.POINTER(Stats), i32p, C.c_int64, ip]
        lib.psd_z_ordschur.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dpp, C.c_char, C.c_int, C.POINTER(C.c_uint8),
                                       C.c_int, dp, dp, i32p, C.POINTER(Stats), ip]
        lib.psd_z_gordschur.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dpp, C.POINTER(C.c_uint8), C.c_char, C.c_int,
                                        C.POINTER(C.c_uint8), C.c_int, dp, dp, i32p, C.POINTER(Stats), ip]
        lib.psd_d_gordschur.argtypes = lib.psd_z_gordschur.argtypes
        lib.psd_d_ordschur.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dpp, C.c_char, C.c_int, C.POINTER(C.c_uint8),
                                       C.c_int, dp, dp, C.POINTER(Stats), ip]
        u8p = C.POINTER(C.c_uint8)
        lib.psd_d_checkpsd.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dpp, dpp, u8p, C.c_char, C.c_int, dp,
                                       C.c_double, C.c_int, dp, dp, dp, ip, ip]
        lib.psd_z_checkpsd.argtypes = [C.c_void_p, C.c_int, C.c_int, dpp, dpp, dpp, u8p, C.c_char, C.c_int,
                                       C.c_double, C.c_int, dp, dp, dp, ip, ip]
        lib.psd_d_checkpsd_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, u8p,
                                           C.c_char, C.c_int, C.c_double, C.c_int, dp, dp, dp, ip, ip]
        lib.psd_d_pschur_hess_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dpp, dpp, C.c_int, C.c_int, C.c_int,
                                                dp, dp, ip, C.POINTER(Stats), ip]
        lib.psd_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.psd_shard_owned.argtypes = [C.c_void_p, C.c_int, C.c_char, u8p]
        self.ctx = C.c_void_p()
        rc = lib.psd_create(C.byref(self.ctx), device)
        if rc != 0:
            raise RuntimeError(f"psd_create failed (info={rc}): no usable HIP device; there is no CPU fallback")

    def version(self):
        return self.lib.psd_version().decode()

    def set_train(self, bulges):
        """Multishift trains of the real pschur! path (psd_set_train): bulges >= 2 (default 32) or 0 for the reference's
        one-shift-one-sweep iteration."""
        self.lib.psd_set_train.argtypes = [C.c_void_p, C.c_int]
        self.lib.psd_set_train(self.ctx, int(bulges))

    def set_train_z(self, bulges):
        self.lib.psd_set_train_z.argtypes = [C.c_void_p, C.c_int]
        self.lib.psd_set_train_z(self.ctx, int(bulges))

    def get_train_z(self):
        self.lib.psd_get_train_z.argtypes = [C.c_void_p]
        return int(self.lib.psd_get_train_z(self.ctx))

    def set_train_g(self, bulges):
        """Multishift trains of the signed paths, real and complex (psd_set_train_g; default 32; 0 = the reference's iteration)."""
        self.lib.psd_set_train_g.argtypes = [C.c_void_p, C.c_int]
        self.lib.psd_set_train_g(self.ctx, int(bulges))

    def get_train_g(self):
        self.lib.psd_get_train_g.argtypes = [C.c_void_p]
        return int(self.lib.psd_get_train_g(self.ctx))

    def get_train(self):
        self.lib.psd_get_train.argtypes = [C.c_void_p]
        return int(self.lib.psd_get_train(self.ctx))

    def set_slices(self, slices):
        """Factor-sliced sweep windows of the real engine (psd_set_slices): `slices` workgroups per window, each with the
        window blocks of its contiguous slice of the period; 1 = off."""
        self.lib.psd_set_slices.argtypes = [C.c_void_p, C.c_int]
        rc = self.lib.psd_set_slices(self.ctx, int(slices))
        if rc != 0:
            raise ValueError(f"psd_set_slices({slices}): argument {-rc} invalid")

    def get_slices(self):
        self.lib.psd_get_slices.argtypes = [C.c_void_p]
        return int(self.lib.psd_get_slices(self.ctx))

    def hess_pipe(self):
        """1 if this engine's multi-stream Hessenberg reductions take the pipe form (psd_get_hess_pipe): fixed at
        creation, forced on by set_shard with world > 1."""
        self.lib.psd_get_hess_pipe.argtypes = [C.c_void_p]
        return int(self.lib.psd_get_hess_pipe(self.ctx))

    def set_shard(self, rank, world):
        """Period sharding (include/psd_mi355x.h, psd_set_shard): this engine keeps the Schur vectors Z_j of its
        contiguous slice of the period; the chains and the factors are computed identically on every rank."""
        rc = self.lib.psd_set_shard(self.ctx, int(rank), int(world))
        if rc != 0:
            raise ValueError(f"psd_set_shard({rank}, {world}): argument {-rc} invalid")
        self.shard = (int(rank), int(world))

    def owned_slots(self, p, lr="R"):
        """Boolean mask over the user slots of Z that this (sharded) engine holds after a call with orientation lr."""
        owned = (C.c_uint8 * p)()
        rc = self.lib.psd_shard_owned(self.ctx, p, char_lr(lr).encode(), owned)
        if rc != 0:
            raise ValueError(f"psd_shard_owned: argument {-rc} invalid")
        return np.array([bool(x) for x in owned])

    def set_profile(self, on):
        self.lib.psd_set_profile(self.ctx, int(on))

    def close(self):
        if self.ctx:
            self.lib.psd_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------------------------------
    @staticmethod
    def _ptrs(mats):
        arr = (C.c_void_p * len(mats))()
        for j, a in enumerate(mats):
            arr[j] = a.ctypes.data
        return arr

    @staticmethod
    def _raise(info):
        if info == 0:
            return
        if info < 0:
            raise ValueError(f"argument {-info} invalid")
        if info >= INFO_RUNTIME:
            raise RuntimeError(f"HIP runtime failure (code {info - INFO_RUNTIME})")
        if info >= INFO_NOTIMPL:
            raise NotImplementedPSD("not implemented in this build")
        if info >= INFO_NOCONV:
            raise ConvergenceError(info - INFO_NOCONV)
        raise RuntimeError(f"info={info}")

    @staticmethod
    def _as_work(A, dtype=np.float64):
        """In-place contract: every A[j] must be a Fortran-contiguous matrix of `dtype` we may overwrite."""
        for a in A:
            if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags.f_contiguous and a.flags.writeable):
                raise TypeError(f"needs writable Fortran-ordered {np.dtype(dtype).name} matrices (use pschur for a copy)")

    @staticmethod
    def _is_complex(A):
        return any(np.iscomplexobj(a) for a in A)

    def phessenberg_(self, A):
        """phessenberg!(A) — src/PeriodicSchurDecompositions.jl:213-259.
        Overwrites A LAPACK-style; returns (H list, tau[p][n]) where H[0] = triu(A[0],-1), H[j] = triu(A[j])."""
        n = _check_square(A)
        if self._is_complex(A):
            self._as_work(A, np.complex128)
            p = len(A)
            tau = np.zeros((p, n), dtype=np.complex128)
            st = Stats()
            info = C.c_int(0)
            self.lib.psd_z_phessenberg(self.ctx, n, p, self._ptrs(A), tau.view(np.float64).ctypes.data_as(
                C.POINTER(C.c_double)), C.byref(st), C.byref(info))
            self._raise(info.value)
            return [np.triu(a, -1 if j == 0 else 0) for j, a in enumerate(A)], tau, st
        self._as_work(A)
        p = len(A)
        tau = np.zeros((p, n))
        st = Stats()
        info = C.c_int(0)
        self.lib.psd_d_phessenberg(self.ctx, n, p, self._ptrs(A), tau.ctypes.data_as(C.POINTER(C.c_double)),
                                   C.byref(st), C.byref(info))
        self._raise(info.value)
        Hs = [np.triu(a, -1 if j == 0 else 0) for j, a in enumerate(A)]
        return Hs, tau, st

    def pschur_(self, A, lr="R", S=None, wantZ=True, wantT=True, maxitfac=30):
        """pschur!(A, lr; wantZ, wantT, maxitfac) — src/PeriodicSchurDecompositions.jl:120-152.
        `A` is workspace and is overwritten with the T factors."""
        orient = char_lr(lr)
        n = _check_square(A)
        if self._is_complex(A):
            return self._zpschur_(A, orient, S, wantZ, wantT, maxitfac)
        self._as_work(A)
        p = len(A)
        if S is not None:  # pschur!(A, S, lr) — src/rgeneralized.jl:3-45 (also for all(S), as the reference)
            return self._gpschur_(A, S, orient, wantZ, wantT, 120 if maxitfac == 30 else maxitfac)
        Z = [np.zeros((n, n), order="F") for _ in range(p)] if wantZ else []
        wr = np.zeros(n)
        wi = np.zeros(n)
        si = C.c_int(0)
        st = Stats()
        maxlog = 2 * maxitfac * n + n + 16
        log = np.zeros(3 * maxlog, dtype=np.int32)
        info = C.c_int(0)
        Sarr = None
        if S is not None:
            if len(S) != p:
                raise DimensionMismatch("length of S must match the period")
            Sarr = (C.c_uint8 * p)(*[1 if s else 0 for s in S])
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_pschur(self.ctx, n, p, self._ptrs(A), Sarr, orient.encode(), int(wantT), int(wantZ), int(maxitfac),
                              self._ptrs(Z) if wantZ else None, wr.ctypes.data_as(dp), wi.ctypes.data_as(dp),
                              C.byref(si), C.byref(st), log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog,
                              C.byref(info))
        self._raise(info.value)
        nl = min(st.nlog, maxlog)
        return PeriodicSchur(list(A), Z, wr + 1j * wi, orient, si.value, st, log[: 3 * nl].reshape(-1, 3).copy())

    def _gpschur_(self, A, S, orient, wantZ, wantT, maxitfac):
        """pschur!(A, S, lr; wantZ, wantT) for Float64 — src/rgeneralized.jl:3-45 -> GeneralizedPeriodicSchur."""
        n = A[0].shape[0]
        p = len(A)
        if len(S) != p:
            raise DimensionMismatch("length of S must match the period")
        first = S[p - 1] if orient == "L" else S[0]
        if not first:
            raise ValueError("The leftmost entry in S must be true")  # src/rgeneralized.jl:37
        Z = [np.zeros((n, n), order="F") for _ in range(p)] if wantZ else []
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        si = C.c_int(0)
        st = Stats()
        info = C.c_int(0)
        Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_gpschur(self.ctx, n, p, self._ptrs(A), Sarr, orient.encode(), int(wantT), int(wantZ),
                               int(maxitfac), self._ptrs(Z) if wantZ else None,
                               alpha.view(np.float64).ctypes.data_as(dp), beta.ctypes.data_as(dp),
                               sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(si), C.byref(st), C.byref(info))
        self._raise(info.value)
        return GeneralizedPeriodicSchur(list(S), list(A), Z, alpha, beta, sc, orient, si.value, st, None)

    def gphessenberg_(self, A, S, wantQ=True):
        """_phessenberg!(A, S; wantQ) for Float64 / ComplexF64 — src/generalized.jl:988-1082.  Overwrites A with the Hessenberg /
        triangular factors; returns (A, Qs)."""
        n = _check_square(A)
        cplx = self._is_complex(A)
        dt = np.complex128 if cplx else np.float64
        self._as_work(A, dt)
        p = len(A)
        if len(S) != p:
            raise DimensionMismatch("length of S must match the period")
        if not S[0]:
            raise ValueError("The leftmost entry in S must be true")  # src/generalized.jl:990
        Q = [np.zeros((n, n), dtype=dt, order="F") for _ in range(p)] if wantQ else []
        st = Stats()
        info = C.c_int(0)
        Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
        fn = self.lib.psd_z_gphessenberg if cplx else self.lib.psd_d_gphessenberg
        fn(self.ctx, n, p, self._ptrs(A), Sarr, self._ptrs(Q) if wantQ else None, C.byref(st), C.byref(info))
        self._raise(info.value)
        self.last_stats = st
        return list(A), Q

    def rphessenberg_(self, Ap, A, Q=None):
        """_rphessenberg!(Ap, A, Q) — src/rhessx.jl:55-109 (Float64 / ComplexF64): row-wise periodic Hessenberg reduction
        for the left orientation.  Ap: m x n with m in (n, n+1); A: p-1 matrices n x n; Q: p matrices with at least n
        columns that are post-multiplied (or None).  Everything is overwritten; returns (Ap, A)."""
        m, n = Ap.shape
        if m not in (n, n + 1):
            raise ValueError("only implemented for square or 1 extra row")  # src/rhessx.jl:62
        p = len(A) + 1
        for a in A:
            if a.shape != (n, n):
                raise DimensionMismatch("all factors must be n x n")  # src/rhessx.jl:66
        cplx = np.iscomplexobj(Ap)
        dt = np.complex128 if cplx else np.float64
        self._as_work([Ap] + list(A) + (list(Q) if Q is not None else []), dt)
        nq = nqc = 0
        if Q is not None:
            if len(Q) != p:
                raise DimensionMismatch("one Q per factor")
            nq, nqc = Q[0].shape
            if nqc < n or any(q.shape != (nq, nqc) for q in Q):
                raise DimensionMismatch("Q matrices must share a shape with at least n columns")
        info = C.c_int(0)
        fn = self.lib.psd_z_rphessenberg if cplx else self.lib.psd_d_rphessenberg
        fn(self.ctx, m, n, p, Ap.ctypes.data, self._ptrs(A) if p > 1 else None, self._ptrs(Q) if Q is not None else None,
           nq, nqc, C.byref(info))
        self._raise(info.value)
        return Ap, list(A)

    def _zpschur_(self, A, orient, S, wantZ, wantT, maxitfac):
        """pschur!(A::Vector{Matrix{ComplexF64}}[, S], lr) — src/PeriodicSchurDecompositions.jl:1106-1111,
        src/generalized.jl:108-148.  Without S the result is a PeriodicSchur (as at :1110)."""
        n = A[0].shape[0]
        self._as_work(A, np.complex128)
        p = len(A)
        Z = [np.zeros((n, n), dtype=np.complex128, order="F") for _ in range(p)] if wantZ else []
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        si = C.c_int(0)
        st = Stats()
        maxlog = 2 * maxitfac * n + n + 16
        log = np.zeros(3 * maxlog, dtype=np.int32)
        info = C.c_int(0)
        Sarr = None
        if S is not None:
            if len(S) != p:
                raise DimensionMismatch("length of S must match the period")
            first = S[p - 1] if orient == "L" else S[0]
            if not first:
                raise ValueError("The leftmost entry in S must be true")  # src/generalized.jl:140
            Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
        dp = C.POINTER(C.c_double)
        self.lib.psd_z_pschur(self.ctx, n, p, self._ptrs(A), Sarr, orient.encode(), int(wantT), int(wantZ),
                              int(maxitfac), self._ptrs(Z) if wantZ else None,
                              alpha.view(np.float64).ctypes.data_as(dp), beta.ctypes.data_as(dp),
                              sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(si), C.byref(st),
                              log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog, C.byref(info))
        self._raise(info.value)
        nl = min(st.nlog, maxlog)
        slog = log[: 3 * nl].reshape(-1, 3).copy()
        g = GeneralizedPeriodicSchur([True] * p if S is None else S, list(A), Z, alpha, beta, sc, orient, si.value, st,
                                     slog)
        if S is None:
            return PeriodicSchur(g.Ts, g.Z, g.values, g.orientation, g.schurindex, st, slog)
        return g

    def pschur(self, A, lr="R", **kw):
        """pschur(A, lr; kwargs...) — copying variant, src/PeriodicSchurDecompositions.jl:108-113."""
        dt = np.complex128 if self._is_complex(A) else np.float64
        Atmp = [np.array(a, dtype=dt, order="F", copy=True) for a in A]
        return self.pschur_(Atmp, lr, **kw)

    def gpschur(self, As, Bs, **kw):
        """gpschur(As, Bs) — src/generalized.jl:1191-1211: generalized periodic Schur decomposition for the formal
        product B_p^-1 A_p ... B_1^-1 A_1 of paired series in left operator order.  As the reference, the arguments are
        complexified and interleaved by `_mkpsargs` (:1198-1211) and handed to pschur!(Cs, Ss)."""
        ph = len(As)
        if len(Bs) != ph:
            raise DimensionMismatch("As and Bs must have the same length")
        cz = lambda m: np.array(m, dtype=np.complex128, order="F", copy=True)  # noqa: E731
        ib = 0 if ph == 1 else ph - 2
        Cs, Ss = [cz(As[ph - 1]), cz(Bs[ib])], [True, False]
        for j in range(ph - 1, 0, -1):  # j = ph-1 .. 1 (1-based)
            Cs.append(cz(As[j - 1]))
            jx = ph if j == 1 else j - 1
            Cs.append(cz(Bs[jx - 1]))
            Ss += [True, False]
        return self.pschur_(Cs, "R", S=Ss, **kw)

    def zpschur_hess_(self, H1, Hs, S=None, Q=None, wantT=True, wantZ=True, maxitfac=30, rev=False):
        """pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac, rev) for ComplexF64 — src/generalized.jl:166-175."""
        H = [H1] + list(Hs)
        n = _check_square(H)
        self._as_work(H, np.complex128)
        p = len(H)
        S = [True] * p if S is None else list(S)
        if not S[0]:
            raise ValueError("Signature entry S[1] must be true")  # src/generalized.jl:182
        if wantZ:
            if Q is None:
                Q = [np.asfortranarray(np.eye(n, dtype=np.complex128)) for _ in range(p)]
            self._as_work(Q, np.complex128)
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        st = Stats()
        maxlog = 2 * maxitfac * n + n + 16
        log = np.zeros(3 * maxlog, dtype=np.int32)
        info = C.c_int(0)
        Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
        dp = C.POINTER(C.c_double)
        self.lib.psd_z_pschur_hess(self.ctx, n, p, self._ptrs(H), Sarr, self._ptrs(Q) if wantZ else None, int(wantT),
                                   int(wantZ), int(maxitfac), alpha.view(np.float64).ctypes.data_as(dp),
                                   beta.ctypes.data_as(dp), sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st),
                                   log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog, C.byref(info))
        self._raise(info.value)
        nl = min(st.nlog, maxlog)
        slog = log[: 3 * nl].reshape(-1, 3).copy()
        Z = list(Q) if wantZ else []
        if rev:  # src/generalized.jl:910-927
            Zr = ([Z[0]] + [Z[p + 1 - l] for l in range(2, p + 1)]) if wantZ else Z
            Ts = [H[p - l] for l in range(1, p)] + [H[0]]
            return GeneralizedPeriodicSchur(S[::-1], Ts, Zr, alpha, beta, sc, "L", p, st, slog)
        return GeneralizedPeriodicSchur(S, H, Z, alpha, beta, sc, "R", 1, st, slog)

    def gpschur_hess_(self, H1, Hs, S, Q=None, wantT=True, wantZ=True, maxitfac=120, rev=False):
        """pschur!(H1, Hs, S; wantT, wantZ, Q, maxitfac, rev) for Float64 — src/rgeneralized.jl:49-59."""
        H = [H1] + list(Hs)
        n = _check_square(H)
        self._as_work(H)
        p = len(H)
        S = list(S)
        if len(S) != p:
            raise DimensionMismatch("S must have one entry per factor")
        if not S[0]:
            raise ValueError("Signature entry S[1] must be true")  # src/rgeneralized.jl:73
        if wantZ:
            if Q is None:
                Q = [np.asfortranarray(np.eye(n)) for _ in range(p)]
            self._as_work(Q)
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        st = Stats()
        maxlog = 2 * maxitfac * n + n + 16
        log = np.zeros(3 * maxlog, dtype=np.int32)
        info = C.c_int(0)
        Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_gpschur_hess(self.ctx, n, p, self._ptrs(H), Sarr, self._ptrs(Q) if wantZ else None, int(wantT),
                                    int(wantZ), int(maxitfac), alpha.view(np.float64).ctypes.data_as(dp),
                                    beta.ctypes.data_as(dp), sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st),
                                    log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog, C.byref(info))
        self._raise(info.value)
        nl = min(st.nlog, maxlog)
        slog = log[: 3 * nl].reshape(-1, 3).copy()
        Z = list(Q) if wantZ else []
        if rev:  # src/rgeneralized.jl:1062-1079
            Zr = ([Z[0]] + [Z[p + 1 - l] for l in range(2, p + 1)]) if wantZ else Z
            Ts = [H[p - l] for l in range(1, p)] + [H[0]]
            return GeneralizedPeriodicSchur(S[::-1], Ts, Zr, alpha, beta, sc, "L", p, st, slog)
        return GeneralizedPeriodicSchur(S, H, Z, alpha, beta, sc, "R", 1, st, slog)

    def pschur_hess_(self, H1, Hs, Q=None, wantT=True, wantZ=True, maxitfac=30, rev=False):
        """pschur!(H1, Hs; wantT, wantZ, Q, maxitfac, rev) — src/PeriodicSchurDecompositions.jl:322-330."""
        H = [H1] + list(Hs)
        n = _check_square(H)
        self._as_work(H)
        p = len(H)
        if wantZ:
            if Q is None:
                Q = [np.asfortranarray(np.eye(n)) for _ in range(p)]
            self._as_work(Q)
        wr = np.zeros(n)
        wi = np.zeros(n)
        st = Stats()
        maxlog = 2 * maxitfac * n + n + 16
        log = np.zeros(3 * maxlog, dtype=np.int32)
        info = C.c_int(0)
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_pschur_hess(self.ctx, n, p, self._ptrs(H), self._ptrs(Q) if wantZ else None, int(wantT),
                                   int(wantZ), int(maxitfac), wr.ctypes.data_as(dp), wi.ctypes.data_as(dp),
                                   C.byref(st), log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog, C.byref(info))
        self._raise(info.value)
        nl = min(st.nlog, maxlog)
        lam = wr + 1j * wi
        Z = list(Q) if wantZ else []
        slog = log[: 3 * nl].reshape(-1, 3).copy()
        if rev:  # src/PeriodicSchurDecompositions.jl:1078-1092
            Zr = ([Z[0]] + [Z[p + 1 - l] for l in range(2, p + 1)]) if wantZ else Z
            Ts = [H[p - l] for l in range(1, p)] + [H[0]]
            return PeriodicSchur(Ts, Zr, lam, "L", p, st, slog)
        return PeriodicSchur(H, Z, lam, "R", 1, st, slog)

    def pschur_hess_batch_(self, problems, wantT=True, wantZ=True, maxitfac=30, infos_out=None):
        """pschur!(H1, Hs; wantT, wantZ, Q, maxitfac) (src/PeriodicSchurDecompositions.jl:322-330) for a list of
        problems of equal shape in ONE call (psd_d_pschur_hess_batch: what src/krylov.jl:575-592,800-829 issues one
        by one).  `problems`: list of (H1, Hs) or (H1, Hs, Q); matrices are overwritten.  Returns a list of
        PeriodicSchur; a problem that fails to converge raises like the single call, after all have run — the other
        problems of the batch are complete then (a failure ends only the problem it occurs in).  `infos_out`: a list
        that receives the per-problem info codes instead (nothing is raised for a failed problem then)."""
        nb = len(problems)
        if nb == 0:
            return []
        Hall, Qall = [], []
        n = problems[0][0].shape[0]
        p = len(problems[0][1]) + 1
        for pr in problems:
            H = [pr[0]] + list(pr[1])
            if len(H) != p or _check_square(H) != n:
                raise DimensionMismatch("the problems of a batch must have equal order and period")
            self._as_work(H)
            Hall += H
            if wantZ:
                Q = list(pr[2]) if len(pr) > 2 and pr[2] is not None else [np.asfortranarray(np.eye(n)) for _ in range(p)]
                self._as_work(Q)
                Qall += Q
        wr = np.zeros((nb, n))
        wi = np.zeros((nb, n))
        infos = (C.c_int * nb)()
        st = Stats()
        info = C.c_int(0)
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_pschur_hess_batch(self.ctx, nb, n, p, self._ptrs(Hall), self._ptrs(Qall) if wantZ else None,
                                         int(wantT), int(wantZ), int(maxitfac), wr.ctypes.data_as(dp),
                                         wi.ctypes.data_as(dp), infos, C.byref(st), C.byref(info))
        if info.value < 0 or info.value >= INFO_NOTIMPL:
            self._raise(info.value)
        out = []
        for q in range(nb):
            Z = Qall[q * p:(q + 1) * p] if wantZ else []
            out.append(PeriodicSchur(Hall[q * p:(q + 1) * p], Z, wr[q] + 1j * wi[q], "R", 1, st))
        if infos_out is not None:
            infos_out[:] = [int(infos[q]) for q in range(nb)]
            return out
        for q in range(nb):
            self._raise(infos[q])
        return out

    def ordschur_(self, P, select, wantZ=True, Z=None):
        """LinearAlgebra.ordschur!(P, select; wantZ, Z) — src/ordschur.jl:11-73 (ComplexF64).  Mutates and returns P.
        `Z`: supplementary matrices that receive the transformations instead of P.Z (src/ordschur.jl:17,34-42; not for
        the right orientation, :36-38)."""
        if Z is not None and wantZ:
            if P.orientation == "R":
                raise NotImplementedPSD("no logic for reversing supplementary Z")  # src/ordschur.jl:37
            if len(Z) != len(P.Ts):
                raise DimensionMismatch("one supplementary Z per factor")
            keep = P.Z
            P.Z = list(Z)
            try:
                return self.ordschur_(P, select, wantZ=True)
            finally:
                P.Z = keep
        n = P.Ts[0].shape[0]
        p = len(P.Ts)
        if len(select) != n:
            raise DimensionMismatch("select must have one entry per eigenvalue")
        if isinstance(P, GeneralizedPeriodicSchur) and not all(P.S):
            return self._gordschur_(P, select, wantZ)
        if not np.iscomplexobj(P.Ts[0]):
            return self._rordschur_(P, select, wantZ)
        if P.schurindex not in (1, p):
            raise ValueError("only implemented for schurindex in (1,p)")  # src/ordschur.jl:32
        self._as_work(P.Ts, np.complex128)
        wantZ = wantZ and len(P.Z) > 0
        if wantZ:
            self._as_work(P.Z, np.complex128)
        sel = (C.c_uint8 * n)(*[1 if x else 0 for x in select])
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        st = Stats()
        info = C.c_int(0)
        dp = C.POINTER(C.c_double)
        self.lib.psd_z_ordschur(self.ctx, n, p, self._ptrs(P.Ts), self._ptrs(P.Z) if wantZ else None,
                                P.orientation.encode(), P.schurindex, sel, int(wantZ),
                                alpha.view(np.float64).ctypes.data_as(dp), beta.ctypes.data_as(dp),
                                sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st), C.byref(info))
        iv = info.value
        if iv == 3000:
            raise SingularException()
        if 2000 <= iv < 3000:
            raise IllConditionedException(iv - 2000)
        self._raise(iv)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            P.values = alpha / beta * np.exp2(sc.astype(np.float64))
        if isinstance(P, GeneralizedPeriodicSchur):
            P.alpha, P.beta, P.alphascale = alpha, beta, sc
        P.stats = st
        return P

    def _gordschur_(self, P, select, wantZ):
        """ordschur!(P::GeneralizedPeriodicSchur, select; wantZ) with a signed S — src/ordschur.jl:11-96,323-328,
        src/sylswap.jl:638-764 (1x1 swaps; a Float64 decomposition needs a real spectrum in this build)."""
        n = P.Ts[0].shape[0]
        p = len(P.Ts)
        if P.schurindex not in (1, p):
            raise ValueError("only implemented for schurindex in (1,p)")  # src/ordschur.jl:32
        cplx = np.iscomplexobj(P.Ts[0])
        dt = np.complex128 if cplx else np.float64
        self._as_work(P.Ts, dt)
        wantZ = wantZ and len(P.Z) > 0
        if wantZ:
            self._as_work(P.Z, dt)
        sel = (C.c_uint8 * n)(*[1 if x else 0 for x in select])
        Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in P.S])
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        st = Stats()
        info = C.c_int(0)
        dp = C.POINTER(C.c_double)
        fn = self.lib.psd_z_gordschur if cplx else self.lib.psd_d_gordschur
        fn(self.ctx, n, p, self._ptrs(P.Ts), self._ptrs(P.Z) if wantZ else None, Sarr, P.orientation.encode(),
           P.schurindex, sel, int(wantZ), alpha.view(np.float64).ctypes.data_as(dp), beta.ctypes.data_as(dp),
           sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(st), C.byref(info))
        iv = info.value
        if iv == 3000:
            raise SingularException()
        if 2000 <= iv < 3000:
            raise IllConditionedException(iv - 2000)
        self._raise(iv)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            P.values = alpha / beta * np.exp2(sc.astype(np.float64))
        P.alpha, P.beta, P.alphascale = alpha, beta, sc
        P.stats = st
        return P

    def _rordschur_(self, P, select, wantZ):
        """LinearAlgebra.ordschur!(P, select; wantZ) for Float64 — src/rordschur.jl:3-132."""
        n = P.Ts[0].shape[0]
        p = len(P.Ts)
        if P.schurindex not in (1, p):
            raise ValueError("only implemented for schurindex in (1,p)")  # src/rordschur.jl:25
        self._as_work(P.Ts)
        wantZ = wantZ and len(P.Z) > 0
        if wantZ:
            self._as_work(P.Z)
        sel = (C.c_uint8 * n)(*[1 if x else 0 for x in select])
        wr = np.zeros(n)
        wi = np.zeros(n)
        st = Stats()
        info = C.c_int(0)
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_ordschur(self.ctx, n, p, self._ptrs(P.Ts), self._ptrs(P.Z) if wantZ else None,
                                P.orientation.encode(), P.schurindex, sel, int(wantZ), wr.ctypes.data_as(dp),
                                wi.ctypes.data_as(dp), C.byref(st), C.byref(info))
        iv = info.value
        if iv == 3000:
            raise SingularException()
        if 2000 <= iv < 3000:
            raise IllConditionedException(iv - 2000)
        self._raise(iv)
        P.values = wr + 1j * wi
        P.stats = st
        return P

    def eigvecs(self, ps0, select, shifted=True):
        """LinearAlgebra.eigvecs(ps::PeriodicSchur, select; shifted) — src/vectors.jl:25-138: selected right
        eigenvectors of the product (and of its circular shifts).  A loop of `ordschur!` calls (on the device) that
        brings one selected eigenvalue (or conjugate pair) after the other to the top, where its vector is read off
        the leading Schur vectors; for a pair the 2x2 cyclic problem is solved (babd.jl, here a dense 2p x 2p solve).
        `select` is completed to conjugate pairs for a real decomposition (vectors.jl:42-62); `ps0` is not modified.
        Returns a list of p (shifted) or one complex n x nvec matrices, normalised so that A_l v_l = mu v_{l+1},
        mu^p = lambda_k (left orientation)."""
        import copy

        if len(ps0.Z) == 0 or ps0.Z[0].shape[0] == 0:
            raise ValueError("eigvecs requires Schur vectors in the PSD")  # vectors.jl:30-32
        n, m = ps0.Z[0].shape
        select = [bool(x) for x in select]
        if len(select) != m:
            raise ValueError("length of `select` must correspond to rank of Schur (sub-)space")  # vectors.jl:34-36
        real = not np.iscomplexobj(ps0.Ts[0])
        dt = np.float64 if real else np.complex128
        ps = PeriodicSchur([np.array(t, dtype=dt, order="F") for t in ps0.Ts],
                           [np.array(z, dtype=dt, order="F") for z in ps0.Z], np.array(ps0.values, dtype=complex),
                           ps0.orientation, ps0.schurindex)
        p = ps.period
        left = ps.orientation == "L"
        if not all(select):
            if real:  # vectors.jl:42-62
                j = 0
                while j < m:
                    if ps.values[j].imag != 0 and j + 1 < m:
                        if select[j] or select[j + 1]:
                            select[j] = select[j + 1] = True
                        j += 2
                    else:
                        j += 1
            self.ordschur_(ps, select)
        nvec = sum(select)
        sel = [k < nvec for k in range(m)]
        nmat = p if shifted else 1
        Vs = [np.zeros((n, nvec), dtype=np.complex128, order="F") for _ in range(nmat)]
        k = 0
        while k < nvec:
            lam = complex(ps.values[0])
            mu = lam ** (1.0 / p)
            if real and lam.imag != 0:
                # the 2x2 cyclic problem (vectors.jl:73-112): | D1 0 .. Lp ; L1 D2 .. ; .. Lp-1 Dp | x = e1 with the
                # first row replaced by the normalisation x_1[1] + x_1[2] = 1
                M = np.zeros((2 * p, 2 * p), dtype=np.complex128)
                for l in range(p):
                    M[2 * l:2 * l + 2, 2 * l:2 * l + 2] += -mu * np.eye(2)
                for l in range(1, p + 1):
                    lx = l if left else (p + 1 - l)
                    blk = ps.Ts[l - 1][0:2, 0:2]
                    r = lx % p  # block column lx (1-based) couples into block row lx + 1 (cyclically)
                    M[2 * r:2 * r + 2, 2 * (lx - 1):2 * (lx - 1) + 2] += blk
                y = np.zeros(2 * p, dtype=np.complex128)
                M[0, :] = 0.0
                M[0, 0:2] = 1.0
                y[0] = 1.0
                x = np.linalg.solve(M, y)
                t = 1.0 / np.linalg.norm(x[0:2])
                for l in range(1, nmat + 1):
                    i0 = (l - 1) * 2 if left else (0 if l == 1 else (p + 1 - l) * 2)
                    Vs[l - 1][:, k] = t * (ps.Z[l - 1][:, 0:2] @ x[i0:i0 + 2])
                    Vs[l - 1][:, k + 1] = np.conj(Vs[l - 1][:, k])
                nl = 2
            else:  # A_1 x_1 = T_1[1,1] Z_2[:,1] = mu x_2, ... (vectors.jl:113-129)
                fac = 1.0 + 0.0j
                for l in range(1, nmat + 1):
                    Vs[l - 1][:, k] = fac * ps.Z[l - 1][:, 0]
                    fac *= ps.Ts[l - 1][0, 0] / mu
                nl = 1
            for q in range(nl):
                sel[q] = False
            self.ordschur_(ps, sel)  # vectors.jl:133
            k += nl
            sel = sel[nl:] + sel[:nl]  # circshift!(sel, -nl)
        return Vs

    def checkpsd(self, P, As, thresh=100, strict=True, S=None, details=False):
        """checkpsd(P, As; thresh, strict) — src/diagnostics.jl:190-263 — evaluated on the device (matrix cores).
        Returns (ok, err) like the reference; with details=True also the orthogonality and triangularity norms."""
        p = len(As)
        if P.period != p:
            raise DimensionMismatch("length of Hs vector must match period of P")  # diagnostics.jl:194-196
        n = P.Ts[0].shape[0]
        for a in As:
            if a.ndim != 2 or a.shape[0] != n or a.shape[1] != n:
                raise DimensionMismatch("size of Hs matrices must match P")  # diagnostics.jl:197-202
        if S is None and isinstance(P, GeneralizedPeriodicSchur):
            S = P.S
        cplx = self._is_complex(P.Ts) or self._is_complex(As) or self._is_complex(P.Z)
        dt = np.complex128 if cplx else np.float64
        Tw = [np.asfortranarray(t, dtype=dt) for t in P.Ts]
        Zw = [np.asfortranarray(z, dtype=dt) for z in P.Z]
        Aw = [np.asfortranarray(a, dtype=dt) for a in As]
        sig = (C.c_uint8 * p)(*[1 if x else 0 for x in S]) if S is not None else None
        err, orth, tri = np.zeros(p), np.zeros(p), np.zeros(p)
        ok, info = C.c_int(0), C.c_int(0)
        dp = C.POINTER(C.c_double)
        if cplx:
            self.lib.psd_z_checkpsd(self.ctx, n, p, self._ptrs(Tw), self._ptrs(Zw), self._ptrs(Aw), sig,
                                    P.orientation.encode(), P.schurindex, float(thresh), int(strict),
                                    err.ctypes.data_as(dp), orth.ctypes.data_as(dp), tri.ctypes.data_as(dp),
                                    C.byref(ok), C.byref(info))
        else:
            wi = np.ascontiguousarray(np.asarray(P.values).imag, dtype=np.float64)
            self.lib.psd_d_checkpsd(self.ctx, n, p, self._ptrs(Tw), self._ptrs(Zw), self._ptrs(Aw), sig,
                                    P.orientation.encode(), P.schurindex, wi.ctypes.data_as(dp), float(thresh),
                                    int(strict), err.ctypes.data_as(dp), orth.ctypes.data_as(dp),
                                    tri.ctypes.data_as(dp), C.byref(ok), C.byref(info))
        self._raise(info.value)
        if details:
            return bool(ok.value), err, orth, tri
        return bool(ok.value), err

    def checkpsd_dev(self, dT_ptr, dZ_ptr, dA_ptr, n, p, lr="R", schurindex=1, thresh=100, strict=True, S=None):
        """checkpsd on operands already resident in HBM ([p][n][n] Float64 blocks in user order)."""
        sig = (C.c_uint8 * p)(*[1 if x else 0 for x in S]) if S is not None else None
        err, orth, tri = np.zeros(p), np.zeros(p), np.zeros(p)
        ok, info = C.c_int(0), C.c_int(0)
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_checkpsd_dev(self.ctx, n, p, C.c_void_p(dT_ptr), C.c_void_p(dZ_ptr), C.c_void_p(dA_ptr), sig,
                                    char_lr(lr).encode(), schurindex, float(thresh), int(strict),
                                    err.ctypes.data_as(dp), orth.ctypes.data_as(dp), tri.ctypes.data_as(dp),
                                    C.byref(ok), C.byref(info))
        self._raise(info.value)
        return bool(ok.value), err, orth, tri

    def pschur_dev(self, dA_ptr, n, p, lr="R", dZ_ptr=None, wantT=True, maxitfac=30):
        """Device-resident pschur!: dA_ptr / dZ_ptr are device addresses of [p][n][n] column-major blocks."""
        orient = char_lr(lr)
        wantZ = dZ_ptr is not None
        wr = np.zeros(n)
        wi = np.zeros(n)
        si = C.c_int(0)
        st = Stats()
        maxlog = 2 * maxitfac * n + n + 16
        log = np.zeros(3 * maxlog, dtype=np.int32)
        info = C.c_int(0)
        dp = C.POINTER(C.c_double)
        self.lib.psd_d_pschur_dev(self.ctx, n, p, C.c_void_p(dA_ptr), orient.encode(), int(wantT), int(wantZ),
                                  int(maxitfac), C.c_void_p(dZ_ptr) if wantZ else None, wr.ctypes.data_as(dp),
                                  wi.ctypes.data_as(dp), C.byref(si), C.byref(st),
                                  log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog, C.byref(info))
        self._raise(info.value)
        nl = min(st.nlog, maxlog)
        return wr + 1j * wi, si.value, st, log[: 3 * nl].reshape(-1, 3).copy()


    def zpschur_dev(self, dA_ptr, n, p, lr="R", dZ_ptr=None, wantT=True, maxitfac=30):
        """Device-resident pschur! for ComplexF64 (psd_z_pschur_dev): dA_ptr / dZ_ptr are device addresses of [p][n][n]
        column-major blocks of interleaved (re, im) doubles.  Returns (values, schurindex, stats, sweep log)."""
        orient = char_lr(lr)
        wantZ = dZ_ptr is not None
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        si = C.c_int(0)
        st = Stats()
        maxlog = 2 * maxitfac * n + n + 16
        log = np.zeros(3 * maxlog, dtype=np.int32)
        info = C.c_int(0)
        dp = C.POINTER(C.c_double)
        i32p = C.POINTER(C.c_int32)
        self.lib.psd_z_pschur_dev(self.ctx, n, p, C.c_void_p(dA_ptr), orient.encode(), int(wantT), int(wantZ),
                                  int(maxitfac), C.c_void_p(dZ_ptr) if wantZ else None, alpha.ctypes.data_as(dp),
                                  beta.ctypes.data_as(dp), sc.ctypes.data_as(i32p), C.byref(si), C.byref(st),
                                  log.ctypes.data_as(i32p), maxlog, C.byref(info))
        self._raise(info.value)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            lam = alpha / beta * np.exp2(sc.astype(np.float64))
        nl = min(st.nlog, maxlog)
        return lam, si.value, st, log[: 3 * nl].reshape(-1, 3).copy()


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine


def pschur(A, lr="R", **kw):
    return default_engine().pschur(A, lr, **kw)


def pschur_(A, lr="R", **kw):
    return default_engine().pschur_(A, lr, **kw)


def phessenberg_(A):
    return default_engine().phessenberg_(A)
