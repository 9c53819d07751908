// Window <-> HBM transfer, in-window rotations and the fused chain link of the real signed periodic QZ.
// Included twice by psd_rgz.h: at global scope (thread index = workgroup thread: the single-wave chase kernel) and in
// namespace psd_wv with the data-parallel macros running over the lanes of ONE wavefront (the multi-wave pipelined
// Hessenberg kernel).  No include guard on purpose.

// factors j0, j0 + jstep, ... (0-based) only.  As psd_win_load (psd_real_qr.h): lane = (row pair, column group of 4),
// two consecutive rows per 16-byte access, two factors per batch; the odd last row travels with the row above it.
PSD_D void psd_gwin_load(const psd_gparams& P, const psd_gwin& w, int n, int p, int j0 = 0, int jstep = 1) {
    const int m = w.be - w.bs + 1;
    PSD_PAR_FOR(t, PSD_STEP_NT) {
        const int r = 2 * (t & 15), g = t >> 4;
        if (r < m) {
            const bool pair = r + 1 < m;
            const int back = (pair || r == 0) ? 0 : 1;
            const bool one = !pair && r == 0;
            for (int j = j0; j < p; j += 2 * jstep) {
                const double* src = P.H + (size_t)j * n * n + (size_t)(w.bs - 1) * n + (w.bs - 1 + r - back);
                double* dst = w.b + j * w.bsz + r;
                psd_pair v[2][8];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = g + 4 * u;
                        const double* q = src + (size_t)f * jstep * n * n + (size_t)c * n;
                        psd_pair x;
                        x.a = x.b = 0.0;
                        if (c < m && j + f * jstep < p) {
                            if (one) x.a = q[0];
                            else x = psd_pair_load(q);
                        }
                        v[f][u] = x;
                    }
                }
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = g + 4 * u;
                        if (c < m && j + f * jstep < p) {
                            double* q = dst + f * jstep * w.bsz + c * w.ld;
                            q[0] = back ? v[f][u].b : v[f][u].a;
                            if (pair) q[1] = v[f][u].b;
                        }
                    }
                }
            }
        }
    }
    PSD_SYNC();
}
PSD_D void psd_gwin_store(const psd_gparams& P, const psd_gwin& w, int n, int p, int j0 = 0, int jstep = 1) {
    const int m = w.be - w.bs + 1;
    PSD_SYNC();
    PSD_PAR_FOR(t, PSD_STEP_NT) {
        const int r = 2 * (t & 15), g = t >> 4;
        if (r < m) {
            const bool pair = r + 1 < m;
            for (int j = j0; j < p; j += jstep) {
                double* dst = P.H + (size_t)j * n * n + (size_t)(w.bs - 1) * n + (w.bs - 1 + r);
                const double* src = w.b + j * w.bsz + r;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = g + 4 * u;
                    if (c < m) {
                        if (pair) {
                            psd_pair x;
                            x.a = src[c * w.ld];
                            x.b = src[c * w.ld + 1];
                            psd_pair_store(dst + (size_t)c * n, x);
                        } else {
                            dst[(size_t)c * n] = src[c * w.ld];
                        }
                    }
                }
            }
        }
    }
    PSD_SYNC();
}

// in-window rmul!(view(H_l, r0:r1, :), G') on columns (j, j+1)
PSD_D void psd_gwin_right(const psd_gwin& w, int l, int j, double c, double s, int r0, int r1) {
    if (r0 < w.bs) r0 = w.bs;
    if (r1 > w.be) r1 = w.be;
    PSD_PAR_FOR(t, r1 - r0 + 1) {
        const int r = r0 + t;
        const double a1 = w.at(l, r, j), a2 = w.at(l, r, j + 1);
        w.at(l, r, j) = c * a1 + s * a2;
        w.at(l, r, j + 1) = c * a2 - s * a1;
    }
    PSD_WAVE_SYNC();
}
// in-window lmul!(G, view(H_l, :, c0:c1)) on rows (j, j+1)
PSD_D void psd_gwin_left(const psd_gwin& w, int l, int j, double c, double s, int c0, int c1) {
    if (c0 < w.bs) c0 = w.bs;
    if (c1 > w.be) c1 = w.be;
    PSD_PAR_FOR(t, c1 - c0 + 1) {
        const int cc = c0 + t;
        const double a1 = w.at(l, j, cc), a2 = w.at(l, j + 1, cc);
        w.at(l, j, cc) = c * a1 + s * a2;
        w.at(l, j + 1, cc) = c * a2 - s * a1;
    }
    PSD_WAVE_SYNC();
}
PSD_D void psd_gwin_set2(const psd_gwin& w, int l, int r1, int c1, double v1, int r2, int c2, double v2) {
    PSD_WAVE_SYNC();
    PSD_ONE {
        w.at(l, r1, c1) = v1;
        w.at(l, r2, c2) = v2;
    }
    PSD_WAVE_SYNC();
}

// One factor of a rotation chain inside the window.  Incoming rotation (c, s) at (q, q+1):
//   cols_in:  it acts on the columns of H_l; the fill H_l[q+1,q] is removed by a new row rotation
//             (rgeneralized.jl:980-991 for S[l] in the downward chain, :922-933 for !S[l] in the forward chain)
//   !cols_in: it acts on the rows; the fill is removed by a new column rotation generated from
//             (H_l[q+1,q+1], -H_l[q+1,q])  (:993-1004, :906-920).  A "backwards" Givens(q+1, q, c, s') of the
//             zero-shift / Case II text (:294-300) is the same rotation.
// Returns the new rotation in (c, s); it acts on the rows (cols_in) or columns (!cols_in) of H_l.
// One fused pass: every operand is loaded once into a lane register (row lanes hold (H[r,q], H[r,q+1]), column
// lanes hold (H[q,cc], H[q+1,cc])), the 2x2 corner travels by v_readlane, one wave-level sync.
// If slot >= 0 the new rotation is also stored as entry `slot` of owner `own`'s list (lane 0, no counter round trip).
PSD_D void psd_g_link(const psd_gwin& w, int l, int q, bool cols_in, double& c, double& s, int rlo, int chi,
                      psd_gtr* trbase = nullptr, int own = 0, int slot = -1) {
    const int r0 = (rlo > w.bs) ? rlo : w.bs;
    const int c1 = (chi < w.be) ? chi : w.be;
    double* base = w.b + (l - 1) * w.bsz;
    PSD_LANEVAR(double, x1);
    PSD_LANEVAR(double, x2);
    PSD_LANEVAR(int, off);
    PSD_LANEVAR(int, str);
    double r;
    if (cols_in) {
        const int nr = q + 2 - r0, nl = c1 - q;  // rows r0..q+1 ; columns q+1..c1
        PSD_PAR_ONCE(t, nr + nl) {
            if (t < nr) {
                PSD_LV(off) = (q - w.bs) * w.ld + (r0 + t - w.bs);
                PSD_LV(str) = w.ld;
            } else {
                PSD_LV(off) = (q + 1 + (t - nr) - w.bs) * w.ld + (q - w.bs);
                PSD_LV(str) = 1;
            }
            const double a1 = base[PSD_LV(off)], a2 = base[PSD_LV(off) + PSD_LV(str)];
            if (t < nr) {
                PSD_LV(x1) = c * a1 + s * a2;
                PSD_LV(x2) = c * a2 - s * a1;
            } else {
                PSD_LV(x1) = a1;
                PSD_LV(x2) = a2;
            }
        }
        const double f = PSD_BCAST(x1, nr - 2), g = PSD_BCAST(x1, nr - 1);
        const double top = PSD_BCAST(x2, nr - 2), bot = PSD_BCAST(x2, nr - 1);
        psd_givens(f, g, c, s, r);
        PSD_PAR_ONCE(t, nr + nl) {
            double* qp = base + PSD_LV(off);
            if (t < nr) {
                if (t >= nr - 2) {  // rows q, q+1: column q becomes (r, 0); their column q+1 belongs to the row pass
                    qp[0] = (t == nr - 2) ? r : 0.0;
                } else {
                    qp[0] = PSD_LV(x1);
                    qp[PSD_LV(str)] = PSD_LV(x2);
                }
            } else {
                const double a1 = (t == nr) ? top : PSD_LV(x1), a2 = (t == nr) ? bot : PSD_LV(x2);
                qp[0] = c * a1 + s * a2;
                qp[1] = c * a2 - s * a1;
            }
        }
    } else {
        const int nl = c1 - q + 1, nr = q - r0;  // columns q..c1 ; rows r0..q-1
        PSD_PAR_ONCE(t, nl + nr) {
            if (t < nl) {
                PSD_LV(off) = (q + t - w.bs) * w.ld + (q - w.bs);
                PSD_LV(str) = 1;
            } else {
                PSD_LV(off) = (q - w.bs) * w.ld + (r0 + (t - nl) - w.bs);
                PSD_LV(str) = w.ld;
            }
            const double a1 = base[PSD_LV(off)], a2 = base[PSD_LV(off) + PSD_LV(str)];
            if (t < nl) {
                PSD_LV(x1) = c * a1 + s * a2;
                PSD_LV(x2) = c * a2 - s * a1;
            } else {
                PSD_LV(x1) = a1;
                PSD_LV(x2) = a2;
            }
        }
        // 2x2 corner after the row rotation: column q in lane 0, column q+1 in lane 1
        const double p00 = PSD_BCAST(x1, 0), p10 = PSD_BCAST(x2, 0), p01 = PSD_BCAST(x1, 1), p11 = PSD_BCAST(x2, 1);
        psd_givens(p11, -p10, c, s, r);
        PSD_PAR_ONCE(t, nl + nr) {
            double* qp = base + PSD_LV(off);
            if (t == 0) {
                qp[0] = c * p00 + s * p01;
                qp[1] = 0.0;
            } else if (t == 1) {
                qp[0] = c * p01 - s * p00;
                qp[1] = r;
            } else if (t < nl) {
                qp[0] = PSD_LV(x1);
                qp[1] = PSD_LV(x2);
            } else {
                const double a1 = PSD_LV(x1), a2 = PSD_LV(x2);
                qp[0] = c * a1 + s * a2;
                qp[PSD_LV(str)] = c * a2 - s * a1;
            }
        }
    }
    if (slot >= 0 && slot < PSD_GTR_CAP) {
        PSD_ONE {
            psd_gtr tr;
            tr.pos = q;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            trbase[(size_t)(own - 1) * PSD_GTR_CAP + slot] = tr;
        }
    }
    PSD_WAVE_SYNC();
}
PSD_D void psd_gstore_tr(const psd_gparams& P, int own, int slot, int pos, double c, double s) {
    if (slot < PSD_GTR_CAP) {
        PSD_ONE {
            psd_gtr tr;
            tr.pos = pos;
            tr.pad = 0;
            tr.c = c;
            tr.s = s;
            P.tr[(size_t)(own - 1) * PSD_GTR_CAP + slot] = tr;
        }
    }
}

