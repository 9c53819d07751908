// Window <-> HBM transfer and in-window rotations of the complex engines.  Included by psd_zqz.h at global scope and
// by psd_zgz.h a second time in namespace psd_wv with the data-parallel macros running over the lanes of one wavefront
// (see psd_rgz_chain.inl).  No include guard on purpose.

// factors j0, j0 + jstep, ... (0-based) only
PSD_D void psd_zwin_load(const psd_zparams& P, const psd_zwin& w, int n, int p, int j0 = 0, int jstep = 1) {
    const int m = w.be - w.bs + 1;
#ifndef PSD_HOSTSIM
    if (PSD_NTHREADS == 64 && w.ld <= 64) {
        // LDS-DMA (global_load_lds_dwordx4, as psd_win_load of the real engine): a lane's element (16 bytes) goes straight
        // to the window image at wave-uniform base + 16 lane, no VGPR destination, so every load of the window is in
        // flight at once; one instruction fills 64 / ld whole columns of one factor (lanes in image order).
        const int cpi = 64 / w.ld;
        const int lane = PSD_TID;
        const int cl = lane / w.ld, r = lane - cl * w.ld;
        const bool on = cl < cpi && r < m;
        const size_t fstride = (size_t)n * n;
        for (int c0 = 0; c0 < m; c0 += cpi) {
            const int c = c0 + cl;
            const bool act = on && c < m;
            const psd_z* q = P.H + (size_t)j0 * fstride + (size_t)(w.bs - 1 + (act ? c : 0)) * n + (w.bs - 1 + (act ? r : 0));
            psd_z* dst = w.b + j0 * w.bsz + c0 * w.ld;
            if (act) {
                for (int j = j0; j < p; j += jstep) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)q,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    q += (size_t)jstep * fstride;
                    dst += jstep * w.bsz;
                }
            }
        }
        PSD_SYNC();
        return;
    }
#endif
    const int RW = (m > 16) ? 32 : 16, sh = (m > 16) ? 5 : 4, ncg = PSD_STEP_NT / RW;
    if (m <= 16) {
        // narrow windows (many factors): 16 rows x 4 column groups, a lane has at most 4 elements per factor, so four
        // factors go into one batch (16 loads in flight) -- a batch costs one memory round trip whatever its size
        PSD_PAR_FOR(t, PSD_STEP_NT) {
            const int r = t & 15, g = t >> 4;
            if (r < m) {
                for (int j = j0; j < p; j += 4 * jstep) {
                    const psd_z* src = P.H + (size_t)j * n * n + (size_t)(w.bs - 1) * n + (w.bs - 1 + r);
                    psd_z* dst = w.b + j * w.bsz + r;
                    psd_z v[4][4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = g + 4 * u;
                            v[f][u] = zmk(0.0, 0.0);
                            if (c < m && j + f * jstep < p) v[f][u] = src[(size_t)f * jstep * n * n + (size_t)c * n];
                        }
                    }
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = g + 4 * u;
                            if (c < m && j + f * jstep < p) dst[f * jstep * w.bsz + c * w.ld] = v[f][u];
                        }
                    }
                }
            }
        }
        PSD_SYNC();
        return;
    }
    PSD_PAR_FOR(t, PSD_STEP_NT) {
        const int r = t & (RW - 1), g = t >> sh;
        if (r < m) {
            for (int j = j0; j < p; j += jstep) {  // (factor loop outside: no index divisions in the hot loop)
                const psd_z* src = P.H + (size_t)j * n * n + (size_t)(w.bs - 1) * n + (w.bs - 1 + r);
                psd_z* dst = w.b + j * w.bsz + r;
                for (int c0 = g; c0 < m; c0 += 8 * ncg) {
                    psd_z v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u * ncg;
                        v[u] = (c < m) ? src[(size_t)c * n] : zmk(0.0, 0.0);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = c0 + u * ncg;
                        if (c < m) dst[c * w.ld] = v[u];
                    }
                }
            }
        }
    }
    PSD_SYNC();
}
PSD_D void psd_zwin_store(const psd_zparams& P, const psd_zwin& w, int n, int p, int j0 = 0, int jstep = 1) {
    const int m = w.be - w.bs + 1;
    const int RW = (m > 16) ? 32 : 16, sh = (m > 16) ? 5 : 4, ncg = PSD_STEP_NT / RW;
    PSD_SYNC();
    PSD_PAR_FOR(t, PSD_STEP_NT) {
        const int r = t & (RW - 1), g = t >> sh;
        if (r < m) {
            for (int j = j0; j < p; j += jstep) {
                psd_z* dst = P.H + (size_t)j * n * n + (size_t)(w.bs - 1) * n + (w.bs - 1 + r);
                const psd_z* src = w.b + j * w.bsz + r;
                for (int c = g; c < m; c += ncg) dst[(size_t)c * n] = src[c * w.ld];
            }
        }
    }
    PSD_SYNC();
}

// in-window: rmul!(view(H_l, r0:r1, :), G') on columns (j, j+1)
PSD_D void psd_zwin_right(const psd_zwin& w, int l, int j, double c, psd_z s, int r0, int r1) {
    if (r0 < w.bs) r0 = w.bs;
    if (r1 > w.be) r1 = w.be;
    PSD_PAR_FOR(t, r1 - r0 + 1) {
        const int r = r0 + t;
        psd_z a1 = w.at(l, r, j), a2 = w.at(l, r, j + 1);
        psd_zrot_right_adj(c, s, a1, a2);
        w.at(l, r, j) = a1;
        w.at(l, r, j + 1) = a2;
    }
    PSD_WAVE_SYNC();
}
// in-window: lmul!(G, view(H_l, :, c0:c1)) on rows (j, j+1)
PSD_D void psd_zwin_left(const psd_zwin& w, int l, int j, double c, psd_z s, int c0, int c1) {
    if (c0 < w.bs) c0 = w.bs;
    if (c1 > w.be) c1 = w.be;
    PSD_PAR_FOR(t, c1 - c0 + 1) {
        const int cc = c0 + t;
        psd_z a1 = w.at(l, j, cc), a2 = w.at(l, j + 1, cc);
        psd_zrot_left(c, s, a1, a2);
        w.at(l, j, cc) = a1;
        w.at(l, j + 1, cc) = a2;
    }
    PSD_WAVE_SYNC();
}

