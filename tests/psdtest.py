"""Shared test infrastructure (NOT product code).

* ctypes loader for the CPU oracle (oracle/libpsd_oracle.so),
* the counter-based synthetic input generator of SURVEY.md §8(d) (splitmix64 -> Box-Muller),
* numpy restatements of the reference's own checkers:
    compare_reigvals   /root/reference/test/testfuncs.jl:28-52
    pschur_check       /root/reference/test/testfuncs.jl:56-145
    checkpsd           /root/reference/src/diagnostics.jl:190-263
  (the decomposition is non-unique, so parity is judged on these invariants, never element-wise).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
EPS = np.finfo(np.float64).eps


# --------------------------------------------------------------------------------------------
# synthetic inputs (identical on every machine; no dependence on numpy's RNG implementation)
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def randn_counter(seed, stream, count):
    """`count` N(0,1) doubles from a counter-based generator: splitmix64(seed, stream, index) -> Box-Muller."""
    with np.errstate(over="ignore"):
        idx = np.arange(count, dtype=np.uint64)
        base = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        u1 = _splitmix64(base + np.uint64(2) * idx)
        u2 = _splitmix64(base + np.uint64(2) * idx + np.uint64(1))
    f1 = ((u1 >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    f2 = ((u2 >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return np.sqrt(-2.0 * np.log(f1)) * np.cos(2.0 * np.pi * f2)


def synth_factors(n, p, seed, dtype=np.float64):
    """p factors n x n with i.i.d. N(0, 1/n) entries (complex: re, im ~ N(0, 1/(2n))): spectral radius ~ 1
    per factor, so the period-p product neither overflows nor underflows."""
    out = []
    for j in range(p):
        if np.issubdtype(dtype, np.complexfloating):
            g = randn_counter(seed, j, 2 * n * n)
            a = (g[: n * n] + 1j * g[n * n:]) / np.sqrt(2.0 * n)
        else:
            a = randn_counter(seed, j, n * n) / np.sqrt(float(n))
        out.append(np.asfortranarray(a.reshape(n, n).astype(dtype)))
    return out


def bench_factors(n, p, seed, dtype=np.float64, eps=0.5):
    """Benchmark / parity ensemble: A_j = I + eps * G_j / sqrt(n), G_j i.i.d. standard normal.

    SURVEY.md §8(d) proposed plain N(0, 1/n) factors, but the period-p product of those has real eigenvalue
    pairs of modulus 1e-16..1e-46 for p >= 16, and on such pairs the reference's own 2x2 standardisation
    (PeriodicSchurDecompositions.jl:900-1054, rotation taken from the explicitly formed product block) leaves a
    sub-diagonal of 1e-11..1e-6 in T1 that is then zeroed (:1066-1073): the reference algorithm fails its own
    `checkpsd` there (measured with the oracle; DESIGN.md "Inputs").  Shifting by I keeps every factor well
    conditioned, so the reference's invariants hold and can be demanded of the GPU path."""
    return [np.asfortranarray(np.eye(n, dtype=dtype) + eps * a) for a in synth_factors(n, p, seed, dtype)]


def rand_uniform_factors(n, p, seed):
    """U(0,1) entries like the reference's own tests (`rand(T,n,n)`, test/runtests.jl:20,93)."""
    out = []
    for j in range(p):
        with np.errstate(over="ignore"):
            u = _splitmix64(_splitmix64(np.uint64(seed) * np.uint64(7919) + np.uint64(j)) + np.arange(n * n, dtype=np.uint64))
        f = (u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        out.append(np.asfortranarray(f.reshape(n, n)))
    return out


def expsplit(p):
    """Kressner's exponentially split example, /root/reference/test/testfuncs.jl:412-421 (literal data)."""
    fac = 0.1
    A1 = np.array([[9, 4, 1, 4, 3, 4], [6, 8, 2, 4, 0, 2], [0, 7, 4, 4, 6, 6], [0, 0, 8, 4, 6, 7],
                   [0, 0, 0, 8, 9, 3], [0, 0, 0, 0, 5, 0]], dtype=np.float64)
    Aj = np.diag([fac, fac ** 2, fac ** 3, 1, 1, 1]).astype(np.float64)
    A = [np.asfortranarray(A1)] + [np.asfortranarray(Aj.copy()) for _ in range(p - 1)]
    lam = [15.6284, -1.31418 - 3.51424j, -1.31418 + 3.51424j, 90 * fac ** p, (1600 / 3) * fac ** (2 * p),
           -(71750 / 11) * fac ** (3 * p)]
    return A, lam


# --------------------------------------------------------------------------------------------
# packing helpers: list of n x n matrices <-> [p][n][n] with column-major blocks
def pack(As, dtype=np.float64):
    p = len(As)
    n = As[0].shape[0]
    out = np.empty((p, n, n), dtype=dtype)
    for j, a in enumerate(As):
        out[j] = np.asarray(a, dtype=dtype).T  # C-order block holding the transpose == column-major matrix
    return out


def unpack(P):
    return [np.asfortranarray(P[j].T.copy()) for j in range(P.shape[0])]


# --------------------------------------------------------------------------------------------
_oracle = None


def oracle_lib():
    """Load (building if needed) the CPU oracle."""
    global _oracle
    if _oracle is not None:
        return _oracle
    so = os.path.join(ORACLE_DIR, "libpsd_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    lib = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    lib.psdo_d_phessenberg.argtypes = [C.c_int, C.c_int, dp, dp]
    lib.psdo_d_hessenberg_q.argtypes = [C.c_int, C.c_int, dp, dp, dp]
    lib.psdo_d_pschur_hess.argtypes = [C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, C.c_int, dp, dp,
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int64)]
    lib.psdo_d_pschur.argtypes = [C.c_int, C.c_int, dp, C.c_char, C.c_int, C.c_int, C.c_int, dp, dp, dp,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int64,
                                  C.POINTER(C.c_int64), dp]
    i32p, i64p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    lib.psdo_z_pschur_hess.argtypes = [C.c_int, C.c_int, dp, u8p, dp, C.c_int, C.c_int, C.c_int, dp, dp, i32p, i64p,
                                       i32p, C.c_int64, i64p]
    lib.psdo_z_phessenberg.argtypes = [C.c_int, C.c_int, dp, dp, dp]
    lib.psdo_z_pschur.argtypes = [C.c_int, C.c_int, dp, C.c_char, C.c_int, C.c_int, C.c_int, dp, dp, dp, i32p,
                                  C.POINTER(C.c_int), i64p, i32p, C.c_int64, i64p, dp]
    lib.psdo_z_ordschur.argtypes = [C.c_int, C.c_int, dp, dp, C.c_char, C.c_int, u8p, C.c_int, dp, dp, i32p, i64p]
    lib.psdo_d_ordschur_real1x1.argtypes = [C.c_int, C.c_int, dp, dp, C.c_char, C.c_int, u8p, C.c_int, dp, i64p]
    lib.psdo_d_ordschur.argtypes = [C.c_int, C.c_int, dp, dp, C.c_char, C.c_int, u8p, C.c_int, dp, dp, i64p]
    lib.psdo_sg_phessenberg.argtypes = [C.c_int, C.c_int, C.c_int, dp, u8p, dp]
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    lib.psdo_gordschur.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, u8p, C.c_char, C.c_int, u8p, C.c_int, dp, dp,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.psdo_d_gpschur_hess.argtypes = [C.c_int, C.c_int, dp, u8p, dp, C.c_int, C.c_int, C.c_int, dp, dp, i32p, i64p]
    lib.psdo_gpschur.argtypes = [C.c_int, C.c_int, C.c_int, dp, u8p, C.c_char, C.c_int, C.c_int, C.c_int, dp, dp, dp,
                                 i32p, C.POINTER(C.c_int), i64p]
    _oracle = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PSD:
    """Result record mirroring the reference's PeriodicSchur (PSD.jl:59-92), with the full user-order T list."""

    def __init__(self, Ts, Zs, values, orientation, schurindex, info=0, niter=0, sweeplog=None, phase_ms=None):
        self.Ts = Ts
        self.Z = Zs
        self.values = values
        self.orientation = orientation
        self.schurindex = schurindex
        self.info = info
        self.niter = niter
        self.sweeplog = sweeplog
        self.phase_ms = phase_ms

    @property
    def T1(self):
        return self.Ts[self.schurindex - 1]

    @property
    def T(self):
        return [t for j, t in enumerate(self.Ts) if j != self.schurindex - 1]

    @property
    def period(self):
        return len(self.Ts)


def oracle_pschur(As, lr="R", wantZ=True, wantT=True, maxitfac=30):
    """CPU restatement of pschur!(A, lr; wantZ, wantT, maxitfac) (PSD.jl:120-152) for Float64."""
    lib = oracle_lib()
    p = len(As)
    n = As[0].shape[0]
    A = pack(As)
    Z = np.zeros((p, n, n))
    wr = np.zeros(n)
    wi = np.zeros(n)
    si = C.c_int(0)
    niter = C.c_int64(0)
    maxlog = 3 * maxitfac * n + 16
    log = np.zeros(3 * maxlog, dtype=np.int32)
    nlog = C.c_int64(0)
    ph = np.zeros(3)
    info = lib.psdo_d_pschur(n, p, _dp(A), lr.encode()[0:1], int(wantT), int(wantZ), maxitfac, _dp(Z), _dp(wr),
                             _dp(wi), C.byref(si), C.byref(niter), log.ctypes.data_as(C.POINTER(C.c_int32)),
                             maxlog, C.byref(nlog), _dp(ph))
    return PSD(unpack(A), unpack(Z) if wantZ else [], wr + 1j * wi, lr, si.value, info, niter.value,
               log[: 3 * nlog.value].reshape(-1, 3).copy(), ph)


def gvalues(alpha, beta, ascale):
    """GeneralizedPeriodicSchur.values (generalized.jl:74-76): alpha ./ beta .* 2 .^ ascale."""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        return alpha / beta * np.exp2(ascale.astype(np.float64))


class GPSD(PSD):
    """Mirror of GeneralizedPeriodicSchur (generalized.jl:31-85)."""

    def __init__(self, S, Ts, Zs, alpha, beta, ascale, orientation, schurindex, info=0, niter=0, sweeplog=None,
                 phase_ms=None):
        super().__init__(Ts, Zs, gvalues(alpha, beta, ascale), orientation, schurindex, info, niter, sweeplog, phase_ms)
        self.S = list(S)
        self.alpha, self.beta, self.ascale = alpha, beta, ascale


def _zp(a):
    return a.view(np.float64).ctypes.data_as(C.POINTER(C.c_double))


def oracle_zpschur(As, lr="R", wantZ=True, wantT=True, maxitfac=30):
    """CPU restatement of pschur!(A::Vector{Matrix{ComplexF64}}, lr) (PSD.jl:1106-1111 -> generalized.jl:108-148)."""
    lib = oracle_lib()
    p = len(As)
    n = As[0].shape[0]
    A = pack(As, np.complex128)
    Z = np.zeros((p, n, n), dtype=np.complex128)
    alpha = np.zeros(n, dtype=np.complex128)
    beta = np.zeros(n)
    sc = np.zeros(n, dtype=np.int32)
    si = C.c_int(0)
    niter = C.c_int64(0)
    maxlog = 3 * maxitfac * n + 16
    log = np.zeros(3 * maxlog, dtype=np.int32)
    nlog = C.c_int64(0)
    ph = np.zeros(3)
    info = lib.psdo_z_pschur(n, p, _zp(A), lr.encode()[0:1], int(wantT), int(wantZ), maxitfac, _zp(Z), _zp(alpha),
                             _dp(beta), sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(si), C.byref(niter),
                             log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog, C.byref(nlog), _dp(ph))
    return GPSD([True] * p, unpack(A), unpack(Z) if wantZ else [], alpha, beta, sc, lr, si.value, info, niter.value,
                log[: 3 * nlog.value].reshape(-1, 3).copy(), ph)


def oracle_zpschur_hess(H1, Hs, S=None, Q=None, wantZ=True, wantT=True, maxitfac=30, rev=False):
    """CPU restatement of pschur!(H1, Hs, S; wantT, wantZ, Q, rev) for ComplexF64 (generalized.jl:166-931)."""
    lib = oracle_lib()
    Hl = [H1] + list(Hs)
    p = len(Hl)
    n = H1.shape[0]
    S = [True] * p if S is None else list(S)
    H = pack(Hl, np.complex128)
    Z = pack(Q if Q is not None else [np.eye(n)] * p, np.complex128)
    alpha = np.zeros(n, dtype=np.complex128)
    beta = np.zeros(n)
    sc = np.zeros(n, dtype=np.int32)
    niter = C.c_int64(0)
    maxlog = 3 * maxitfac * n + 16
    log = np.zeros(3 * maxlog, dtype=np.int32)
    nlog = C.c_int64(0)
    Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
    info = lib.psdo_z_pschur_hess(n, p, _zp(H), Sarr, _zp(Z), int(wantT), int(wantZ), maxitfac, _zp(alpha), _dp(beta),
                                  sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(niter),
                                  log.ctypes.data_as(C.POINTER(C.c_int32)), maxlog, C.byref(nlog))
    Ts, Zs = unpack(H), (unpack(Z) if wantZ else [])
    slog = log[: 3 * nlog.value].reshape(-1, 3).copy()
    if rev:  # generalized.jl:910-927
        Zr = ([Zs[0]] + [Zs[p + 1 - l] for l in range(2, p + 1)]) if wantZ else Zs
        Tr = [Ts[p - l] for l in range(1, p)] + [Ts[0]]
        return GPSD(S[::-1], Tr, Zr, alpha, beta, sc, "L", p, info, niter.value, slog)
    return GPSD(S, Ts, Zs, alpha, beta, sc, "R", 1, info, niter.value, slog)


def oracle_zphessenberg(As):
    lib = oracle_lib()
    p = len(As)
    n = As[0].shape[0]
    A = pack(As, np.complex128)
    tau = np.zeros((p, n), dtype=np.complex128)
    Q = np.zeros((p, n, n), dtype=np.complex128)
    lib.psdo_z_phessenberg(n, p, _zp(A), _zp(tau), _zp(Q))
    packed = unpack(A)
    Hs = [np.triu(a, -1 if j == 0 else 0) for j, a in enumerate(packed)]
    return Hs, unpack(Q), packed, tau


def gpschur_check(As, S, ps, qtol=10, tol=100):
    """test/testfuncs.jl:155-235 (complex, non-developing branch)."""
    p = len(S)
    n = As[0].shape[0]
    left = ps.orientation == "L"
    Ts, Zs = ps.Ts, ps.Z
    out = {"resid": [], "orth": []}
    for l in range(p):
        ln = (l + 1) % p
        if bool(S[l]) != left:
            Ax = Zs[l] @ Ts[l] @ Zs[ln].conj().T
        else:
            Ax = Zs[ln] @ Ts[l] @ Zs[l].conj().T
        assert np.all(np.tril(Ts[l], -2) == 0)  # istriu(Ts[l], -1)
        orth = np.linalg.norm(Zs[l] @ Zs[l].conj().T - np.eye(n))
        assert orth < qtol * EPS * n, f"Z[{l+1}] orthogonality {orth:.3e}"
        res = np.linalg.norm(As[l] - Ax)
        assert res < tol * EPS * n * max(1.0, np.linalg.norm(As[l], 1) / n), f"residual[{l+1}] {res:.3e}"
        out["orth"].append(orth / (EPS * n))
        out["resid"].append(res / (EPS * n))
    # eigenvalues consistent with the diagonals of the Schur factors (testfuncs.jl:211-234)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        lam_s = np.ones(n, dtype=complex)
        for l in range(p):
            d = np.diag(Ts[l])
            lam_s = lam_s * (d if ps.S[l] else 1.0 / d)
    for j in range(n):
        if np.isfinite(lam_s[j]):
            assert np.isclose(ps.values[j], lam_s[j], rtol=1e-8, atol=0), (j, ps.values[j], lam_s[j])
        else:
            assert not np.isfinite(ps.values[j])
    return out


RG_COUNTERS = ("niter", "sweeps", "zero", "case2", "case3", "real2x2", "cplx2x2", "iwarn")


def oracle_gpschur_hess(H1, Hs, S, Q=None, wantZ=True, wantT=True, maxitfac=120, rev=False):
    """CPU restatement of pschur!(H1, Hs, S; wantT, wantZ, Q, rev) for Float64 (rgeneralized.jl:49-1083)."""
    lib = oracle_lib()
    Hl = [H1] + list(Hs)
    p = len(Hl)
    n = H1.shape[0]
    S = list(S)
    H = pack(Hl, np.float64)
    Z = pack(Q if Q is not None else [np.eye(n)] * p, np.float64)
    alpha = np.zeros(n, dtype=np.complex128)
    beta = np.zeros(n)
    sc = np.zeros(n, dtype=np.int32)
    cnt = np.zeros(8, dtype=np.int64)
    Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
    info = lib.psdo_d_gpschur_hess(n, p, _dp(H), Sarr, _dp(Z), int(wantT), int(wantZ), maxitfac, _zp(alpha), _dp(beta),
                                   sc.ctypes.data_as(C.POINTER(C.c_int32)), cnt.ctypes.data_as(C.POINTER(C.c_int64)))
    Ts, Zs = unpack(H), (unpack(Z) if wantZ else [])
    if rev:  # rgeneralized.jl:1062-1079
        Zr = ([Zs[0]] + [Zs[p + 1 - l] for l in range(2, p + 1)]) if wantZ else Zs
        Tr = [Ts[p - l] for l in range(1, p)] + [Ts[0]]
        out = GPSD(S[::-1], Tr, Zr, alpha, beta, sc, "L", p, info, int(cnt[0]))
    else:
        out = GPSD(S, Ts, Zs, alpha, beta, sc, "R", 1, info, int(cnt[0]))
    out.counters = dict(zip(RG_COUNTERS, cnt.tolist()))
    return out


def oracle_gpschur(As, S, lr="R", wantZ=True, wantT=True, maxitfac=None):
    """CPU restatement of pschur!(A, S, lr) — Float64: rgeneralized.jl:3-45, ComplexF64: generalized.jl:108-148."""
    lib = oracle_lib()
    p = len(As)
    n = As[0].shape[0]
    cplx = any(np.iscomplexobj(a) for a in As)
    dt = np.complex128 if cplx else np.float64
    if maxitfac is None:
        maxitfac = 30 if cplx else 120
    A = pack(As, dt)
    Z = np.zeros((p, n, n), dtype=dt)
    alpha = np.zeros(n, dtype=np.complex128)
    beta = np.zeros(n)
    sc = np.zeros(n, dtype=np.int32)
    cnt = np.zeros(8, dtype=np.int64)
    si = C.c_int(0)
    Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
    ptr = _zp if cplx else _dp
    info = lib.psdo_gpschur(n, p, int(cplx), ptr(A), Sarr, lr.encode()[0:1], int(wantT), int(wantZ), maxitfac, ptr(Z),
                            _zp(alpha), _dp(beta), sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(si),
                            cnt.ctypes.data_as(C.POINTER(C.c_int64)))
    out = GPSD(list(S), unpack(A), unpack(Z) if wantZ else [], alpha, beta, sc, lr, si.value, info, int(cnt[0]))
    out.counters = dict(zip(RG_COUNTERS, cnt.tolist()))
    return out


def rgpschur_check(As, S, ps, qtol=10, tol=100, lam_check=True):
    """test/testfuncs.jl:238-382 (real eltype, non-developing branch) plus, when every eigenvalue is finite, a
    comparison of the spectrum with LAPACK's for the explicitly formed product."""
    p = len(S)
    n = As[0].shape[0]
    left = ps.orientation == "L"
    Ts, Zs = ps.Ts, ps.Z
    js = ps.schurindex - 1
    for l in range(p):
        ln = (l + 1) % p
        if bool(S[l]) != left:
            Ax = Zs[l] @ Ts[l] @ Zs[ln].T
        else:
            Ax = Zs[ln] @ Ts[l] @ Zs[l].T
        if l == js:
            for i in range(n - 1):
                if ps.values[i].imag == 0:
                    assert Ts[l][i + 1, i] == 0, (i, Ts[l][i + 1, i])
        assert np.all(np.tril(Ts[l], -2 if l == js else -1) == 0)
        orth = np.linalg.norm(Zs[l] @ Zs[l].T - np.eye(n))
        assert orth < qtol * EPS * n, f"Z[{l+1}] orthogonality {orth:.3e}"
        res = np.linalg.norm(As[l] - Ax)
        assert res < tol * EPS * n * max(1.0, np.linalg.norm(As[l], 1) / n), f"residual[{l+1}] {res:.3e}"
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        lam_s = np.ones(n)
        for l in range(p):
            d = np.diag(Ts[l])
            lam_s = lam_s * (d if ps.S[l] else 1.0 / d)
    for j in range(n):
        if ps.values[j].imag != 0:
            continue
        if np.isfinite(lam_s[j]):
            assert np.isclose(ps.values[j].real, lam_s[j], rtol=1e-8, atol=0), (j, ps.values[j], lam_s[j])
        else:
            assert not np.isfinite(ps.values[j])
    if lam_check and np.all(np.isfinite(ps.values)):
        P = np.eye(n)
        for l in range(p):
            F = As[l] if S[l] else np.linalg.inv(As[l])
            P = F @ P if left else P @ F
        ref = np.linalg.eigvals(P)
        scale = np.linalg.cond(P) * EPS * np.linalg.norm(P, 2)
        lam = np.array(ps.values)
        used = np.zeros(n, dtype=bool)
        for z in lam:
            d = np.abs(ref - z) + used * 1e300
            k = int(np.argmin(d))
            assert d[k] <= max(1e-8 * abs(z), 100 * scale), (z, ref[k], d[k], scale)
            used[k] = True


def oracle_gordschur(ps, select, wantZ=True):
    """CPU restatement of ordschur!(P::GeneralizedPeriodicSchur, select) by 1x1 swaps (ordschur.jl:11-96,
    sylswap.jl:638-764).  Returns a new GPSD record and the number of swaps."""
    lib = oracle_lib()
    p = len(ps.Ts)
    n = ps.Ts[0].shape[0]
    cplx = np.iscomplexobj(ps.Ts[0])
    dt = np.complex128 if cplx else np.float64
    T = pack(ps.Ts, dt)
    Z = pack(ps.Z, dt) if wantZ else None
    ptr = _zp if cplx else _dp
    sel = (C.c_uint8 * n)(*[1 if x else 0 for x in select])
    Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in ps.S])
    alpha = np.zeros(n, dtype=np.complex128)
    beta = np.zeros(n)
    sc = np.zeros(n, dtype=np.int32)
    nsw = C.c_int64(0)
    info = lib.psdo_gordschur(n, p, int(cplx), ptr(T), ptr(Z) if wantZ else None, Sarr, ps.orientation.encode()[0:1],
                              ps.schurindex, sel, int(wantZ), _zp(alpha), _dp(beta),
                              sc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nsw))
    out = GPSD(ps.S, unpack(T), unpack(Z) if wantZ else [], alpha, beta, sc, ps.orientation, ps.schurindex, info)
    out.nswaps = nsw.value
    return out


def gord_test_factors(n, p, S, seed, cplx):
    """test/ordschur.jl:167-190: well separated real spectrum 4^j spread over the factors with signature S, hidden by
    random unitary equivalences."""
    rng = np.random.default_rng(seed)
    def rnd(shape):
        x = rng.random(shape)
        return x + 1j * rng.random(shape) if cplx else x
    A = [0.01 * np.triu(rnd((n, n))) for _ in range(p)]
    for j in range(n):
        # the reference's n = 7 spectrum 4^j; for larger n the same range [4, 4^7] is kept (well conditioned factors)
        mu = 2.0 ** (2 * (j + 1) / p) if n <= 7 else 2.0 ** (2 * (1 + 6 * j / (n - 1)) / p)
        for l in range(p):
            A[l][j, j] = mu if S[l] else 1.0 / mu
    for l in range(p):
        g = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0)
        q, _ = np.linalg.qr(g)
        A[l] = q @ A[l] if S[l] else A[l] @ q.conj().T
        l1 = (l + 1) % p
        A[l1] = A[l1] @ q.conj().T if S[l1] else q @ A[l1]
    return [np.asfortranarray(a) for a in A]


def rand_uniform_zfactors(n, p, seed):
    re = rand_uniform_factors(n, p, seed)
    im = rand_uniform_factors(n, p, seed + 100003)
    return [np.asfortranarray(a + 1j * b) for a, b in zip(re, im)]


def oracle_ordschur(ps, select, wantZ=True):
    """CPU restatement of ordschur!(P, select) (ordschur.jl:11-73; real: rordschur.jl:3-132 for 1x1 blocks only).
    Returns a new PSD record (inputs are not modified) and the number of adjacent swaps."""
    lib = oracle_lib()
    p = len(ps.Ts)
    n = ps.Ts[0].shape[0]
    cplx = np.iscomplexobj(ps.Ts[0])
    dt = np.complex128 if cplx else np.float64
    T = pack(ps.Ts, dt)
    Z = pack(ps.Z, dt) if wantZ else None
    sel = (C.c_uint8 * n)(*[1 if x else 0 for x in select])
    nsw = C.c_int64(0)
    if cplx:
        alpha = np.zeros(n, dtype=np.complex128)
        beta = np.zeros(n)
        sc = np.zeros(n, dtype=np.int32)
        info = lib.psdo_z_ordschur(n, p, _zp(T), _zp(Z) if wantZ else None, ps.orientation.encode()[0:1], ps.schurindex,
                                   sel, int(wantZ), _zp(alpha), _dp(beta), sc.ctypes.data_as(C.POINTER(C.c_int32)),
                                   C.byref(nsw))
        vals = gvalues(alpha, beta, sc)
    else:
        wr = np.zeros(n)
        wi = np.zeros(n)
        info = lib.psdo_d_ordschur(n, p, _dp(T), _dp(Z) if wantZ else None, ps.orientation.encode()[0:1],
                                   ps.schurindex, sel, int(wantZ), _dp(wr), _dp(wi), C.byref(nsw))
        vals = wr + 1j * wi
    out = PSD(unpack(T), unpack(Z) if wantZ else [], vals, ps.orientation, ps.schurindex, info)
    out.nswaps = nsw.value
    return out


def ord_test_factors(n, p, seed, dtype=np.float64):
    """test/ordschur.jl:4-20: constructed spectrum lambda_j = 4^j (diag mu_j = 2^(2j/p) in every factor, 0.01*U(0,1)
    strict upper part), hidden by a cycle of orthogonal similarity transformations."""
    cplx = np.issubdtype(dtype, np.complexfloating)
    A = []
    for l in range(p):
        u = rand_uniform_zfactors(n, 1, seed + 11 * l)[0] if cplx else rand_uniform_factors(n, 1, seed + 11 * l)[0]
        a = 0.01 * np.triu(u)
        for j in range(n):
            a[j, j] = 2.0 ** (2 * (j + 1) / p)
        A.append(np.asfortranarray(a.astype(dtype)))
    for l in range(p):
        g = randn_counter(seed + 999, l, n * n).reshape(n, n)
        if cplx:
            g = g + 1j * randn_counter(seed + 998, l, n * n).reshape(n, n)
        q, _ = np.linalg.qr(g)
        A[l] = np.asfortranarray(q @ A[l])
        l1 = (l + 1) % p
        A[l1] = np.asfortranarray(A[l1] @ q.conj().T)
    return A


def mkrps(n, p, jcs, seed, nnfac=1e-2, alt=False):
    """test/ordschur.jl:62-125 `mkrps` (alt = false, tri = false): a left-oriented real periodic Schur decomposition
    with schurindex p and conjugate pairs 4^jj (1 +- i) in the (1-based) rows jcs, jcs+1.  Returns (PSD, As)."""
    T1 = np.triu(nnfac * rand_uniform_factors(n, 1, seed)[0])
    Ts = [np.triu(nnfac * rand_uniform_factors(n, 1, seed + 1 + l)[0]) for l in range(p - 1)]
    lam = np.zeros(n, dtype=complex)
    jj = 0
    mu = 0.0
    for j in range(1, n + 1):
        if (j - 1) in jcs:
            T1[j - 1, j - 2] = mu
            T1[j - 2, j - 1] = -mu
            lam[j - 1] = 2.0 ** (2 * jj) * (1 - 1j)
            lam[j - 2] = 2.0 ** (2 * jj) * (1 + 1j)
            for l in range(p - 1):
                Ts[l][j - 2, j - 1] = 0
        else:
            jj += 1
            mu = 2.0 ** (2 * jj / p)
            lam[j - 1] = 2.0 ** (2 * jj)
        T1[j - 1, j - 1] = mu
        for l in range(p - 1):
            Ts[l][j - 1, j - 1] = mu
    q1, _ = np.linalg.qr(randn_counter(seed + 500, 0, n * n).reshape(n, n))
    q2, _ = np.linalg.qr(randn_counter(seed + 500, 1, n * n).reshape(n, n))
    Zs = [q1] + [q2.copy() for _ in range(p - 1)]
    if p > 1:
        As = [Zs[l + 1] @ Ts[l] @ Zs[l].T for l in range(p - 1)] + [Zs[0] @ T1 @ Zs[p - 1].T]
    else:
        As = [Zs[0] @ T1 @ Zs[0].T]
    if alt:  # test/ordschur.jl:108-118: every other factor enters inverted
        S = [True] * p
        for l in range(0, p - 1, 2):
            S[l] = False
            Ts[l] = np.linalg.inv(Ts[l])
            As[l] = Zs[l] @ Ts[l] @ Zs[l + 1].T
        full = [np.asfortranarray(t) for t in Ts] + [np.asfortranarray(T1)]
        ps = GPSD(S, full, [np.asfortranarray(z) for z in Zs], lam.copy(), np.ones(n), np.zeros(n, dtype=np.int32), "L", p)
        return ps, [np.asfortranarray(a) for a in As]
    full = [np.asfortranarray(t) for t in Ts] + [np.asfortranarray(T1)]
    ps = PSD(full, [np.asfortranarray(z) for z in Zs], lam, "L", p)
    return ps, [np.asfortranarray(a) for a in As]


def oracle_sg_phessenberg(As, S):
    """CPU restatement of _phessenberg!(A, S) (generalized.jl:988-1082): returns (Hs, Qs)."""
    lib = oracle_lib()
    p = len(As)
    n = As[0].shape[0]
    cplx = any(np.iscomplexobj(a) for a in As)
    dt = np.complex128 if cplx else np.float64
    A = pack(As, dt)
    Q = np.zeros((p, n, n), dtype=dt)
    Sarr = (C.c_uint8 * p)(*[1 if x else 0 for x in S])
    ptr = _zp if cplx else _dp
    info = lib.psdo_sg_phessenberg(n, p, int(cplx), ptr(A), Sarr, ptr(Q))
    assert info == 0
    return unpack(A), unpack(Q)


def sg_hess_check(A, S, Hs, Qs, tol=20, qtol=10):
    """test/generalized.jl:16-36."""
    p = len(A)
    n = A[0].shape[0]
    assert np.all(np.tril(Hs[0], -2) == 0)
    for j in range(p):
        if j > 0:
            assert np.all(np.tril(Hs[j], -1) == 0)
        assert np.linalg.norm(Qs[j] @ Qs[j].conj().T - np.eye(n)) < qtol * EPS * n
        jn = (j + 1) % p
        Ax = Qs[j] @ Hs[j] @ Qs[jn].conj().T if S[j] else Qs[jn] @ Hs[j] @ Qs[j].conj().T
        assert np.linalg.norm(A[j] - Ax) < tol * EPS * n * max(1.0, np.linalg.norm(A[j], 1) / n), (j, np.linalg.norm(A[j] - Ax))


def oracle_phessenberg(As):
    lib = oracle_lib()
    p = len(As)
    n = As[0].shape[0]
    A = pack(As)
    tau = np.zeros((p, n))
    lib.psdo_d_phessenberg(n, p, _dp(A), _dp(tau))
    Q = np.zeros((p, n, n))
    lib.psdo_d_hessenberg_q(n, p, _dp(A), _dp(tau), _dp(Q))
    packed = unpack(A)
    Hs = [np.triu(a, -1 if j == 0 else 0) for j, a in enumerate(packed)]
    return Hs, unpack(Q), packed, tau


# --------------------------------------------------------------------------------------------
# checkers
def product(As, left=False):
    P = np.eye(As[0].shape[0], dtype=As[0].dtype)
    for a in As:
        P = a @ P if left else P @ a
    return P


def compare_reigvals(lam, lamx, ltol):
    """test/testfuncs.jl:28-52: sort by modulus, real / conjugate-pair aware; returns max scaled error.
    Raises AssertionError on a real/complex mismatch or |dlam| >= ltol * |lam|max."""
    lam = np.asarray(lam, dtype=complex)
    lamx = np.asarray(lamx, dtype=complex)
    n = len(lam)
    idx = np.argsort(np.abs(lam), kind="stable")
    idxx = np.argsort(np.abs(lamx), kind="stable")
    scale = abs(lam[idx[-1]])
    worst = 0.0
    i = 0
    while i < n:
        l1, l1x = lam[idx[i]], lamx[idxx[i]]
        if l1.imag == 0:
            assert l1x.imag == 0, f"eigenvalue {i}: reference real {l1}, got complex {l1x}"
            worst = max(worst, abs(l1 - l1x))
            i += 1
        else:
            l2, l2x = lam[idx[i + 1]], lamx[idxx[i + 1]]
            if l1.imag * l1x.imag < 0:
                l1x, l2x = l2x, l1x
            worst = max(worst, abs(l1 - l1x), abs(l2 - l2x))
            i += 2
    assert worst < ltol * scale, f"eigenvalue mismatch {worst:.3e} >= {ltol:.3e} * {scale:.3e}"
    return worst / scale if scale > 0 else 0.0


def match_eigs(lam, lamx):
    """Greedy multiset match (robust to near-equal moduli); returns max |dlam|."""
    lam = list(np.asarray(lam, dtype=complex))
    lamx = list(np.asarray(lamx, dtype=complex))
    worst = 0.0
    for l in lam:
        d = [abs(l - x) for x in lamx]
        k = int(np.argmin(d))
        worst = max(worst, d[k])
        lamx.pop(k)
    return worst


def pschur_check(As, ps, qtol=10, tol=32, ltol=1000, check_lam=True, lam=None, real=True):
    """test/testfuncs.jl:56-145 (non-developing branch). Returns dict of measured quantities."""
    p = len(As)
    n = As[0].shape[0]
    left = ps.orientation == "L"
    js = ps.schurindex
    Ts, Zs = ps.Ts, ps.Z
    out = {"resid": [], "orth": []}
    for j in range(p):
        jn = (j + 1) % p
        Ax = Zs[jn] @ Ts[j] @ Zs[j].conj().T if left else Zs[j] @ Ts[j] @ Zs[jn].conj().T
        k = -1 if (j == js - 1) else 0
        assert np.all(np.tril(Ts[j], k - 1) == 0), f"T[{j+1}] not (quasi-)triangular"
        if j == js - 1 and real:
            for i in range(n - 1):
                if ps.values[i].imag == 0:
                    assert Ts[j][i + 1, i] == 0, f"subdiagonal junk at {i+1} for real eigenvalue"
        orth = np.linalg.norm(Zs[j] @ Zs[j].conj().T - np.eye(n))
        assert orth < qtol * EPS * n, f"Z[{j+1}] orthogonality {orth:.3e}"
        res = np.linalg.norm(As[j] - Ax)
        bound = tol * EPS * np.linalg.norm(As[j], 1)
        assert res < bound, f"residual[{j+1}] {res:.3e} >= {bound:.3e}"
        out["orth"].append(orth / (EPS * n))
        out["resid"].append(res / (EPS * np.linalg.norm(As[j], 1)))
    if check_lam:
        if lam is None:
            lam = np.linalg.eigvals(product(As, left))
        out["lam_err"] = compare_reigvals(lam, ps.values, ltol * EPS)
    return out


def checkpsd(ps, As, thresh=100, strict=True, S=None):
    """src/diagnostics.jl:190-263: returns (ok, err[]) with err = ||Z T Z' - A|| / eps / ||A||_1."""
    p = len(As)
    n = As[0].shape[0]
    S = [True] * p if S is None else S
    ok = True
    err = np.zeros(p)
    real = not np.iscomplexobj(ps.Ts[0])
    for l in range(p):
        l1 = (l + 1) % p
        Tl = ps.Ts[l]
        cmp_ = 0 if strict else 10 * EPS * n
        tval = np.tril(Tl, -2) if (real and l == ps.schurindex - 1) else np.tril(Tl, -1)
        if np.linalg.norm(tval) > cmp_:
            ok = False
        if np.linalg.norm(ps.Z[l] @ ps.Z[l].conj().T - np.eye(n)) > 10 * EPS * n:
            ok = False
        if S[l] != (ps.orientation == "L"):
            Hx = ps.Z[l] @ Tl @ ps.Z[l1].conj().T
        else:
            Hx = ps.Z[l1] @ Tl @ ps.Z[l].conj().T
        err[l] = np.linalg.norm(Hx - As[l]) / EPS / np.linalg.norm(As[l], 1)
        if err[l] > thresh:
            ok = False
    return ok, err


# ------------------------------------------------------------------------------------------------
# _rphessenberg! (rhessx.jl:55-109): row-wise periodic Hessenberg reduction, left orientation
def oracle_rphessenberg(Ap, As, Qs=None):
    """CPU restatement.  Ap: m x n (m = n or n + 1); As: p-1 matrices n x n; Qs: p matrices nq x nqc (nqc >= n) that
    are post-multiplied, or None.  Returns (Ap, As, Qs) reduced."""
    lib = oracle_lib()
    m, n = Ap.shape
    p = len(As) + 1
    cplx = np.iscomplexobj(Ap) or any(np.iscomplexobj(a) for a in As)
    dt = np.complex128 if cplx else np.float64
    ptr = _zp if cplx else _dp
    apw = np.asfortranarray(Ap.astype(dt))
    A = pack(As, dt) if p > 1 else np.zeros((1, 1, 1), dtype=dt)
    if Qs is not None:
        nq, nqc = Qs[0].shape
        Q = pack(Qs, dt)
        qp = ptr(Q)
    else:
        nq = nqc = 0
        Q = None
        qp = None
    fa = apw.reshape(-1, order="F").copy()
    info = lib.psdo_rphessenberg(m, n, p, int(cplx), ptr(fa), ptr(A), qp, nq, nqc)
    assert info == 0
    return fa.reshape((m, n), order="F"), (unpack(A) if p > 1 else []), (unpack(Q) if Q is not None else None)


def rphess_check(Ap0, As0, Ap, As, Qs, tol=40):
    """Invariants of the reduction started with Q_l = I_n: with U_l = Q_l (orthogonal),
    A_l^new = U_{l+1}' A_l U_l (l < p), Ap^new = blockdiag(U_1, 1)' Ap U_p; Ap upper Hessenberg (an extra row keeps its
    last entry only), A_l upper triangular."""
    m, n = Ap0.shape
    p = len(As0) + 1
    assert np.all(np.tril(Ap, -2) == 0)
    for l in range(p - 1):
        assert np.all(np.tril(As[l], -1) == 0)
    for l in range(p):
        assert np.linalg.norm(Qs[l].conj().T @ Qs[l] - np.eye(n)) < tol * EPS * n
    U1 = np.eye(m, dtype=Ap.dtype)
    U1[:n, :n] = Qs[0]
    sc = max(1.0, np.linalg.norm(Ap0, 1))
    assert np.linalg.norm(U1.conj().T @ Ap0 @ Qs[p - 1] - Ap) < tol * EPS * n * sc, np.linalg.norm(U1.conj().T @ Ap0 @ Qs[p - 1] - Ap)
    for l in range(p - 1):
        sc = max(1.0, np.linalg.norm(As0[l], 1))
        R = Qs[l + 1].conj().T @ As0[l] @ Qs[l] - As[l]
        assert np.linalg.norm(R) < tol * EPS * n * sc, (l, np.linalg.norm(R))
