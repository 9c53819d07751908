"""CPU rehearsal of the north_star multi-GPU partition: the periodic QZ/QR
sweep with the FACTORS sharded by period and the chain handed from rank to rank at the slice boundaries.

Not the product and not a fallback: numpy + torch.distributed (gloo) test infrastructure that pins down the data flow the
peer-mapped-flag kernel of DESIGN.md section 7 implements on xGMI, and proves on world sizes 2 and 3 that it yields a
valid periodic Schur decomposition with every rank holding ONLY the H_j and Z_j of its slice.

Algorithm: the ComplexF64 single-shift periodic QR of the reference (BASELINE configs[2]: pschur! for ComplexF64;
/root/reference/src/generalized.jl:808-852 sweep, :786-805 shift, :260-278 deflation test 1, :741-762 1x1 split), all
signatures true, on a Hessenberg-triangular input.  A transformation generated at factor j acts on the rows of H_j, the
columns of H_{j-1 (cyclic)} and the columns of Z_j (generalized.jl:826-844) — so with rank g owning the contiguous
factors [lo_g, hi_g) everything but ONE rotation per (position, slice boundary) is local:

  * a lap of the bulge at position k: rank 0 (owner of H_1) makes G_1 on rows (k, k+1) of H_1 and hands (c, s) to rank
    G-1, the owner of H_p, whose columns it acts on; each rank applies the incoming rotation to the columns of its top
    factor, works down through its own factors and hands its last rotation to the rank below; rank 0 closes the lap on
    the columns of H_1 and starts the next one.  Record: (bulge, position, c, Re s, Im s) = 32 bytes with the tag.
  * >= G bulges are in flight (a multishift train; shifts = eigenvalues of the trailing block of the product): bulge b
    enters 3 G + 1 beats after bulge b - 1, so at every beat each rank holds at most one bulge, neighbouring bulges stay
    three positions apart, and in steady state every rank works in every beat.  One beat = one rank's share of one lap.
  * what needs all the factors is scalar: the trailing block of the product for the shifts and the diagonal products
    for the start rotation and the eigenvalues — an associative product of small triangular blocks: each rank multiplies
    its slice, ONE all-gather of the G partials, every rank combines them.
  * deflation: test 1 looks at H_1 only (rank 0), the verdict is broadcast.
"""
import numpy as np

EPS = np.finfo(float).eps


def period_slice(p, world, rank):
    base, rem = divmod(p, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _rot(f, g):
    """(c, s) with [c s; -conj(s) c] [f; g] = [r; 0], c real (LAPACK zlartg convention, as givensAlgorithm)."""
    if g == 0:
        return 1.0, 0.0 + 0.0j
    if f == 0:
        return 0.0, np.conj(g) / abs(g)
    d = np.hypot(abs(f), abs(g))
    c = abs(f) / d
    s = (f / abs(f)) * np.conj(g) / d
    return c, s


def _left(M, k, c, s, c0=0):
    """rows (k, k+1) of M <- G rows, from column c0 on"""
    a, b = M[k, c0:].copy(), M[k + 1, c0:].copy()
    M[k, c0:] = c * a + s * b
    M[k + 1, c0:] = -np.conj(s) * a + c * b


def _right_adj(M, k, c, s, r1):
    """columns (k, k+1) of M <- M G^H, rows 0 .. r1 - 1"""
    a, b = M[:r1, k].copy(), M[:r1, k + 1].copy()
    M[:r1, k] = c * a + np.conj(s) * b
    M[:r1, k + 1] = -s * a + c * b


class Ring:
    """dist: an initialised torch.distributed (gloo) module or None (one rank)."""

    def __init__(self, dist=None):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.p2p_messages = 0
        self.p2p_bytes = 0
        self.collectives = 0

    def allgather(self, arr):
        """all-gather of one small complex array per rank -> list by rank"""
        if self.dist is None:
            return [arr]
        import torch

        self.collectives += 1
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, dtype=np.complex128)))
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.numpy() for o in out]

    def bcast0(self, arr):
        if self.dist is None:
            return arr
        import torch

        self.collectives += 1
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, dtype=np.float64)))
        self.dist.broadcast(t, src=0)
        return t.numpy()

    def ring_step(self, recs):
        """every rank hands its records ([m][4] doubles: bulge tag, c, Re s, Im s; tag 0 = none) to the rank below
        (0 -> world - 1) and takes the ones from above: point to point, no collective"""
        if self.dist is None:
            return recs
        import torch

        dst, src = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        out = torch.from_numpy(np.ascontiguousarray(recs))
        inn = torch.zeros_like(out)
        ops = [self.dist.P2POp(self.dist.isend, out, dst), self.dist.P2POp(self.dist.irecv, inn, src)]
        for r in self.dist.batch_isend_irecv(ops):
            r.wait()
        k = int((recs[:, 0] != 0).sum())
        self.p2p_messages += k
        self.p2p_bytes += 32 * k
        return inn.numpy()


def ring_pschur_hess(Hloc, Zloc, n, p, ring, bulges=None, maxit=40):
    """In place: the factors Hloc[j - lo] (j in this rank's slice [lo, hi) of 0..p-1; factor 0 = H_1 upper Hessenberg,
    the others upper triangular) become T_j, Zloc[j - lo] is post-multiplied by the transformations.  Returns
    (eigenvalues, statistics).  Every rank returns the same eigenvalues."""
    G, g = ring.world, ring.rank
    lo, hi = period_slice(p, G, g)
    own0 = lo == 0  # this rank owns H_1 (rank 0)
    m = bulges if bulges else max(2 * G, 4)
    stats = {"trains": 0, "sweeps": 0, "beats": 0, "busy_beats": 0}
    ilast = n - 1  # 0-based index of the last row of the active block
    its = 0

    def diag_products(skip_h1):
        """prod_j H_j[q, q] for all q (optionally without H_1): slice partials + one all-gather"""
        part = np.ones(n, dtype=np.complex128)
        for j in range(max(lo, 1) if skip_h1 else lo, hi):
            part = part * np.diag(Hloc[j - lo])
        out = np.ones(n, dtype=np.complex128)
        for a in ring.allgather(part):
            out = out * a
        return out

    def through(k, c, s, jtop, jbot):
        """the lap at position k through this rank's factors jtop .. jbot (descending): incoming rotation on the columns,
        the factor's own on its rows and on Z_j; returns the rotation that leaves"""
        for j in range(jtop, jbot - 1, -1):
            Hj = Hloc[j - lo]
            _right_adj(Hj, k, c, s, min(k + 2, n))
            c, s = _rot(Hj[k, k], Hj[k + 1, k])
            _left(Hj, k, c, s, k)
            Hj[k + 1, k] = 0.0
            _right_adj(Zloc[j - lo], k, c, s, n)
        return c, s

    while ilast >= 0:
        # ---- deflation search (generalized.jl:260-278, test 1: H_1 alone) on rank 0, verdict to everyone
        info = np.zeros(2)
        if own0:
            H1 = Hloc[0]
            for k in range(ilast, 0, -1):
                if abs(H1[k, k - 1]) <= EPS * (abs(H1[k - 1, k - 1]) + abs(H1[k, k])):
                    H1[k, k - 1] = 0.0
                    info[0] = k
                    break
        ifirst = int(ring.bcast0(info)[0])
        if ifirst == ilast:  # 1x1 block: an eigenvalue (generalized.jl:741-762)
            ilast -= 1
            its = 0
            continue
        its += 1
        if its > maxit:
            raise RuntimeError("ring rehearsal: no convergence")
        w = ilast - ifirst + 1
        mm = min(m, max(1, (w - 1) // 3))  # bulges three positions apart must fit the block
        # ---- shifts: eigenvalues of the trailing mm x mm block of the product H_1 H_2 ... H_p.  The triangular factors'
        # trailing (mm + 1) x (mm + 1) blocks multiply as blocks: slice partials, one all-gather, ordered combine.
        t0 = ilast - mm
        bs = mm + 1
        part = np.eye(bs, dtype=np.complex128)
        for j in range(max(lo, 1), hi):
            part = part @ Hloc[j - lo][t0:ilast + 1, t0:ilast + 1]
        h1blk = Hloc[0][t0:ilast + 1, t0:ilast + 1].copy() if own0 else np.zeros((bs, bs), dtype=np.complex128)
        gathered = ring.allgather(np.concatenate([part.ravel(), h1blk.ravel()]))
        T = np.eye(bs, dtype=np.complex128)
        for a in gathered:
            T = T @ a[: bs * bs].reshape(bs, bs)
        H1b = gathered[0][bs * bs:].reshape(bs, bs)
        Pt = (H1b @ T)[1:, 1:]  # (rows t0 + 1 .. of H_1 have no entry left of column t0: exact)
        shifts = np.linalg.eigvals(Pt)
        shifts = shifts[np.argsort(abs(shifts - Pt[-1, -1]))]  # (the one closest to the corner entry first)
        D = diag_products(True)
        # ---- the train: systolic schedule.  Bulge b enters at beat b (3 G + 1); a lap is G beats: in beat t0_b + G q + e the
        # bulge is at position ifirst + q, on rank 0 for e = 0 and on rank G - e for e >= 1.  Rank 0's share closes the lap
        # before (through its own triangular factors, then on the columns of H_1) and starts the next one in the SAME beat.
        stats["trains"] += 1
        stats["sweeps"] += mm
        gap = 3 * G + 1
        nlaps = ilast - ifirst  # positions ifirst .. ilast - 1
        total_beats = (mm - 1) * gap + G * nlaps + 1
        rec_in = np.zeros((mm, 4))
        for beat in range(total_beats):
            rec_out = np.zeros((mm, 4))
            busy = False
            for b in range(mm):  # (earlier bulges are further down the block: they go first)
                tb = beat - b * gap
                if tb < 0 or tb > G * nlaps:
                    continue
                q, e = divmod(tb, G)
                if (0 if e == 0 else G - e) != g:
                    continue
                busy = True
                k = ifirst + q
                c = s = None
                if tb > 0:
                    assert int(rec_in[b, 0]) == b + 1, (beat, b, rec_in[b])  # the record that arrived is this bulge's
                    c, s = rec_in[b, 1], rec_in[b, 2] + 1j * rec_in[b, 3]
                if e == 0:
                    if q > 0:  # close lap q - 1 (position k - 1)
                        c, s = through(k - 1, c, s, hi - 1, 1)
                        _right_adj(Hloc[0], k - 1, c, s, min(k + 2, n))
                    if q < nlaps:  # start lap q
                        H1 = Hloc[0]
                        if q == 0:
                            c, s = _rot(H1[k, k] - shifts[b % len(shifts)] / D[k], H1[k + 1, k])
                        else:
                            c, s = _rot(H1[k, k - 1], H1[k + 1, k - 1])
                        _left(H1, k, c, s, max(k - 1, 0))
                        if q > 0:
                            H1[k + 1, k - 1] = 0.0
                        _right_adj(Zloc[0], k, c, s, n)
                        rec_out[b] = [b + 1, c, s.real, s.imag]
                else:
                    c, s = through(k, c, s, hi - 1, lo)
                    rec_out[b] = [b + 1, c, s.real, s.imag]
            stats["beats"] += 1
            stats["busy_beats"] += 1 if busy else 0
            rec_in = ring.ring_step(rec_out)
    return diag_products(False), stats
