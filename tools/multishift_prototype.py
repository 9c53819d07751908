#!/usr/bin/env python3
"""Convergence study for the planned multishift periodic QR sweep — numpy only, no GPU.

A train of m double-shift bulges chased back to back is, in exact arithmetic, the same as m consecutive sweeps whose
shift pairs were all fixed BEFORE the first of them (the interleaved transforms commute).  So the question "how many
bulge laps does the multishift iteration need compared with the reference's one-sweep-one-shift iteration" can be
answered with a dense simulation: periodic Hessenberg-triangular form, standard double-shift sweeps (PSD.jl:768-886 in
dense form), m shift pairs per train taken from the eigenvalues of the trailing 2m x 2m block of the product
(m = 1: the reference's strategy), deflation by the usual negligible-subdiagonal test on the explicit product.

usage: multishift_prototype.py [n] [p] [seed]
"""
import sys

import numpy as np

EPS = np.finfo(float).eps


def house(x):
    """dlarfg-style reflector: H x = beta e1, H = I - tau [1;v][1;v]'"""
    x = np.asarray(x, dtype=float)
    alpha, tail = x[0], x[1:]
    xn = np.linalg.norm(tail)
    if xn == 0.0:
        return np.zeros(len(x) - 1), 0.0
    beta = -np.copysign(np.hypot(alpha, xn), alpha)
    return tail / (alpha - beta), (beta - alpha) / beta


def left(M, r0, v, tau, c0=0):
    u = np.concatenate(([1.0], v))
    blk = M[r0:r0 + len(u), c0:]
    blk -= tau * np.outer(u, u @ blk)


def right(M, c0, v, tau, r1=None):
    u = np.concatenate(([1.0], v))
    blk = M[:r1, c0:c0 + len(u)]
    blk -= tau * np.outer(blk @ u, u)


def hess_tri(A):
    """periodic Hessenberg-triangular reduction (PSD.jl:213-259, dense): H_1 Hessenberg, H_j upper triangular"""
    H = [a.copy() for a in A]
    p, n = len(H), H[0].shape[0]
    for i in range(n - 1):
        for j in range(p - 1, 0, -1):
            v, tau = house(H[j][i:, i])
            left(H[j], i, v, tau)
            right(H[j - 1], i, v, tau)
        if i + 1 < n - 1 or True:
            v, tau = house(H[0][i + 1:, i])
            if len(v):
                left(H[0], i + 1, v, tau)
                right(H[p - 1], i + 1, v, tau)
    for j in range(1, p):
        H[j] = np.triu(H[j])
    H[0] = np.triu(H[0], -1)
    return H


def product(H, lo, hi):
    P = H[0][lo:hi + 1, lo:hi + 1].copy()
    for j in range(1, len(H)):
        P = P @ H[j][lo:hi + 1, lo:hi + 1]
    return P


def sweep(H, l, i, tr, det):
    """one double-shift sweep on the active block l..i (0-based, inclusive) with shift pair (sum tr, product det)"""
    p = len(H)
    P = product(H, l, min(l + 2, i))
    x = np.array([P[0, 0] * P[0, 0] + P[0, 1] * P[1, 0] - tr * P[0, 0] + det,
                  P[1, 0] * (P[0, 0] + P[1, 1] - tr),
                  P[2, 1] * P[1, 0] if P.shape[0] > 2 else 0.0])
    for k in range(l, i):
        nr = min(3, i - k + 1)
        if k > l:
            x = H[0][k:k + nr, k - 1].copy()
        v, tau = house(x[:nr])
        left(H[0], k, v, tau, max(k - 1, 0) if k > l else 0)
        right(H[p - 1], k, v, tau, min(k + nr + 1, i + 1))
        for j in range(p - 1, 0, -1):
            v, tau = house(H[j][k:k + nr, k])
            left(H[j], k, v, tau, k)
            right(H[j - 1], k, v, tau, min(k + nr + 1, i + 1))
            if nr == 3:  # restore the triangular form of H_j in column k+1 (PSD.jl:865-881)
                v2, tau2 = house(H[j][k + 1:k + 3, k + 1])
                left(H[j], k + 1, v2, tau2, k + 1)
                right(H[j - 1], k + 1, v2, tau2, min(k + nr + 1, i + 1))


def solve(A, m, maxtrains=10000):
    """returns (bulge laps, trains, bulge positions, train positions) until every eigenvalue is deflated; m shift pairs per
    train.  bulge positions = sum over laps of the active width (the serial work); train positions = sum over trains of
    (active width + 4 (bulges - 1)) (the time of a train whose bulges run side by side, 4 positions apart)"""
    H = hess_tri(A)
    n = H[0].shape[0]
    i = n - 1
    laps = trains = 0
    bpos = tpos = 0
    while i >= 1 and trains < maxtrains:
        # deflation search from the bottom on the explicit product band
        P = product(H, 0, i)
        l = 0
        for k in range(i, 0, -1):
            if abs(P[k, k - 1]) <= EPS * (abs(P[k, k]) + abs(P[k - 1, k - 1])):
                l = k
                break
        if l == i:
            i -= 1
            continue
        if l == i - 1:
            i -= 2
            continue
        if l > 0:
            H[0][l, l - 1] = 0.0
        w = i - l + 1
        mm = max(1, min(m, w // 4))  # a train needs room: 4 positions per bulge
        T = product(H, max(i - 2 * mm + 1, l), i)
        ev = np.linalg.eigvals(T)
        ev = ev[np.argsort(-np.abs(ev.imag))]
        pairs = []
        cplx = [z for z in ev if z.imag > 0]
        real = sorted([z.real for z in ev if z.imag == 0])
        for z in cplx:
            pairs.append((2 * z.real, abs(z) ** 2))
        for q in range(0, len(real) - 1, 2):
            pairs.append((real[q] + real[q + 1], real[q] * real[q + 1]))
        pairs = pairs[:mm] if pairs else [(2 * ev[0].real, abs(ev[0]) ** 2)]
        # bulges enter in order of increasing distance from the trailing 2x2 block's shifts (closest last)
        for (tr, det) in pairs:
            sweep(H, l, i, tr, det)
            laps += 1
            bpos += i - l
        trains += 1
        tpos += (i - l) + 4 * (len(pairs) - 1)
    return laps, trains, bpos, tpos


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    p = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rs = np.random.RandomState(seed)
    A = [np.eye(n) + 0.5 * rs.randn(n, n) / np.sqrt(n) for _ in range(p)]
    ref = np.sort_complex(np.linalg.eigvals(np.linalg.multi_dot(A) if p > 1 else A[0]))
    for m in (1, 2, 4, 8):
        laps, trains, bpos, tpos = solve(A, m)
        print(f"n={n} p={p} seed={seed}  m={m}: {laps} laps in {trains} trains, {bpos} bulge positions "
              f"({bpos / (n * n):.2f} n^2), train time {tpos} positions ({tpos / (n * n):.2f} n^2)")
