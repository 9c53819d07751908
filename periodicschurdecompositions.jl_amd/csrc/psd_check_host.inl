// Host side of the device verifier (psd_check.h): checkpsd(P, As; thresh, strict), src/diagnostics.jl:190-263.
// Included at the end of psd_engine.cpp (one translation unit).

namespace {

// device allocation freed on every way out of the function that owns it
struct psd_devbuf {
    void* p = nullptr;
    ~psd_devbuf() {
        if (p) psd_rt_free(p);
    }
    int alloc(size_t bytes) { return psd_rt_malloc(&p, bytes); }
    double* d() const { return (double*)p; }
};

// T, Z, A: [p][n][n] device blocks in USER order (ES doubles per element).  err[p] (host) receives the normalized
// factorization errors, orth[p] / tri[p] (host, optional) the orthogonality and triangularity norms.
template <bool CPLX>
int checkpsd_dev(psd_ctx* c, int n, int p, const double* dT, const double* dZ, const double* dA, const uint8_t* S,
                 char orient, int schurindex, const double* dwi, double thresh, int strict, double* err, double* orth,
                 double* tri, int* ok) {
    constexpr int E = CPLX ? 2 : 1;
    const size_t nn = (size_t)n * n * E;
    psd_devbuf bW, bacc;
    PSD_CHECK(bW.alloc(nn * sizeof(double)));
    PSD_CHECK(bacc.alloc(sizeof(double) * 5 * (size_t)p));
    double *dW = bW.d(), *dacc = bacc.d();
    PSD_CHECK(psd_rt_memset(dacc, 0, sizeof(double) * 5 * (size_t)p, c->stream));
    const bool left = orient == 'L';
    const int tiles = (n + PSD_CK_TM - 1) / PSD_CK_TM;
    (void)tiles;
    for (int l = 0; l < p; ++l) {
        const int l1 = (l + 1) % p;
        const double* Tl = dT + (size_t)l * nn;
        const double* Al = dA + (size_t)l * nn;
        const int sub = (!CPLX && l == schurindex - 1) ? 1 : 0;  // diagnostics.jl:223-233
        if (CPLX)
            PSD_LAUNCH(psd_ck_small_z, psd_dim3(1), 256, 256 * 8, c->stream, Tl, Al, n, sub, dwi, dacc + 5 * l + 2);
        else
            PSD_LAUNCH(psd_ck_small_d, psd_dim3(1), 256, 256 * 8, c->stream, Tl, Al, n, sub, dwi, dacc + 5 * l + 2);
        const bool sl = S ? (S[l] != 0) : true;
        const int a = (sl != left) ? l : l1, b = (sl != left) ? l1 : l;  // diagnostics.jl:247-251
        psd_ck_args g[3];
        g[0].A = dZ + (size_t)l * nn; g[0].B = dZ + (size_t)l * nn; g[0].C = nullptr; g[0].D = nullptr;
        g[0].acc = dacc + 5 * l + 0; g[0].n = n; g[0].mode = 2; g[0].bconjt = 1;
        g[1].A = Tl; g[1].B = dZ + (size_t)b * nn; g[1].C = dW; g[1].D = nullptr;
        g[1].acc = nullptr; g[1].n = n; g[1].mode = 0; g[1].bconjt = 1;
        g[2].A = dZ + (size_t)a * nn; g[2].B = dW; g[2].C = nullptr; g[2].D = Al;
        g[2].acc = dacc + 5 * l + 1; g[2].n = n; g[2].mode = 1; g[2].bconjt = 0;
        for (int q = 0; q < 3; ++q) {
#ifdef PSD_HOSTSIM
            psd_ck_gemm_sim<CPLX>(g[q]);
#else
            hipLaunchKernelGGL(psd_ck_gemm<CPLX>, dim3(tiles, tiles), dim3(256), 0, c->stream, g[q]);
#endif
        }
    }
    std::vector<double> acc(5 * (size_t)p);
    PSD_CHECK(psd_rt_d2h(acc.data(), dacc, sizeof(double) * 5 * (size_t)p, c->stream));
    PSD_CHECK(psd_rt_sync(c->stream));
    PSD_CHECK(psd_rt_last_error());
    const double eps = PSD_DBL_EPS;
    bool good = true;
    for (int l = 0; l < p; ++l) {
        const double o = sqrt(acc[5 * l + 0]), r = sqrt(acc[5 * l + 1]), t = sqrt(acc[5 * l + 2]);
        const double anorm = acc[5 * l + 3];
        const double cmp = strict ? 0.0 : 10.0 * eps * n;
        if (t > cmp) good = false;
        if (o > 10.0 * eps * n) good = false;
        const double e = r / eps / anorm;
        if (!(e <= thresh)) good = false;
        err[l] = e;
        if (orth) orth[l] = o;
        if (tri) tri[l] = t;
    }
    if (ok) *ok = good ? 1 : 0;
    return 0;
}

template <bool CPLX>
int checkpsd_host(psd_ctx* c, int n, int p, double* const* T, double* const* Z, double* const* A, const uint8_t* S,
                  char orient, int schurindex, const double* wi, double thresh, int strict, double* err, double* orth,
                  double* tri, int* ok, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!T || !Z || !A) return *info = -4;
    if (orient != 'R' && orient != 'L') return *info = -8;
    if (schurindex < 1 || schurindex > p) return *info = -9;
    if (!err) return *info = -13;
    constexpr int E = CPLX ? 2 : 1;
    const size_t nn = (size_t)n * n * E;
    psd_devbuf bT, bZ, bA, bwi;
    PSD_CHECK(bT.alloc(nn * p * sizeof(double)));
    PSD_CHECK(bZ.alloc(nn * p * sizeof(double)));
    PSD_CHECK(bA.alloc(nn * p * sizeof(double)));
    double *dT = bT.d(), *dZ = bZ.d(), *dA = bA.d(), *dwi = nullptr;
    for (int l = 0; l < p; ++l) {
        PSD_CHECK(psd_rt_h2d(dT + l * nn, T[l], nn * 8, c->stream));
        PSD_CHECK(psd_rt_h2d(dZ + l * nn, Z[l], nn * 8, c->stream));
        PSD_CHECK(psd_rt_h2d(dA + l * nn, A[l], nn * 8, c->stream));
    }
    if (wi && !CPLX) {
        PSD_CHECK(bwi.alloc(sizeof(double) * (size_t)n));
        dwi = bwi.d();
        PSD_CHECK(psd_rt_h2d(dwi, wi, sizeof(double) * (size_t)n, c->stream));
    }
    const int rc = checkpsd_dev<CPLX>(c, n, p, dT, dZ, dA, S, orient, schurindex, dwi, thresh, strict, err, orth, tri, ok);
    return *info = rc;
}

}  // namespace

extern "C" {

int psd_d_checkpsd(psd_ctx* c, int n, int p, double* const* T, double* const* Z, double* const* A, const uint8_t* S,
                   char orient, int schurindex, const double* wi, double thresh, int strict, double* err, double* orth,
                   double* tri, int* ok, int* info) {
    return checkpsd_host<false>(c, n, p, T, Z, A, S, orient, schurindex, wi, thresh, strict, err, orth, tri, ok, info);
}

int psd_z_checkpsd(psd_ctx* c, int n, int p, double* const* T, double* const* Z, double* const* A, const uint8_t* S,
                   char orient, int schurindex, double thresh, int strict, double* err, double* orth, double* tri, int* ok,
                   int* info) {
    return checkpsd_host<true>(c, n, p, T, Z, A, S, orient, schurindex, nullptr, thresh, strict, err, orth, tri, ok, info);
}

int psd_d_checkpsd_dev(psd_ctx* c, int n, int p, const double* dT, const double* dZ, const double* dA, const uint8_t* S,
                       char orient, int schurindex, double thresh, int strict, double* err, double* orth, double* tri,
                       int* ok, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (!c) return *info = -1;
    if ((*info = check_dims(n, p)) != 0) return *info;
    if (!dT || !dZ || !dA) return *info = -4;
    if (orient != 'R' && orient != 'L') return *info = -8;
    if (schurindex < 1 || schurindex > p) return *info = -9;
    if (!err) return *info = -12;
    return *info = checkpsd_dev<false>(c, n, p, dT, dZ, dA, S, orient, schurindex, nullptr, thresh, strict, err, orth, tri, ok);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Batch of small Hessenberg-triangular problems in ONE call (the projected problems of the
// Krylov driver, krylov.jl:575-592,800-829, are <= 40 x 40 — one such problem per launch chain is the worst case for a
// launch-bound design).  The slot scheduler of the real iteration (psd_rq_step_mb) runs the problems side by side: each
// starts as one range with its own leader, its own factors, band arrays, eigenvalues and sweep budget; splits and
// trains take further slots as in a single problem.
extern "C" {
int psd_d_pschur_hess_batch(psd_ctx* c, int nb, int n, int p, double* const* H, double* const* Q, int wantT, int wantZ,
                            int maxitfac, double* wr, double* wi, int* infos, psd_stats* stats, int* info) {
    int dummy;
    if (!info) info = &dummy;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!c) return *info = -1;
    if (nb < 1 || nb > PSD_SLOTS / 2) return *info = -2;
    if ((*info = check_dims(n, p)) != 0) return *info - 1;
    if (!H) return *info = -5;
    if (wantZ && !Q) return *info = -6;
    if (maxitfac < 1) return *info = -9;
    if (!wr || !wi) return *info = -10;
    const int mlog = 2 * maxitfac * n * nb + n * nb + 16;
    if ((*info = c->reserve(n, p, false, mlog)) != 0) return *info;
    const size_t nn = (size_t)n * n, sb = (size_t)nb * (n + 8);
    double *dH = nullptr, *dZ = nullptr, *bws = nullptr;
    PSD_CHECK(psd_rt_malloc((void**)&dH, nn * p * nb * sizeof(double)));
    if (wantZ) PSD_CHECK(psd_rt_malloc((void**)&dZ, nn * p * nb * sizeof(double)));
    PSD_CHECK(psd_rt_malloc((void**)&bws, (8 * sb + (size_t)nb * (p + 8)) * sizeof(double)));
    PSD_CHECK(psd_rt_memset(bws, 0, (8 * sb + (size_t)nb * (p + 8)) * sizeof(double), c->stream));
    for (int q = 0; q < nb; ++q)
        for (int j = 0; j < p; ++j) {
            PSD_CHECK(psd_rt_h2d(dH + ((size_t)q * p + j) * nn, H[(size_t)q * p + j], nn * 8, c->stream));
            if (wantZ) PSD_CHECK(psd_rt_h2d(dZ + ((size_t)q * p + j) * nn, Q[(size_t)q * p + j], nn * 8, c->stream));
        }
    psd_stats local;
    memset(&local, 0, sizeof(local));
    psd_stats* s = stats ? stats : &local;
    Timer t;
    t.start(c->stream);
    psd_rstate st;
    std::vector<int> pinfo(nb, 0);
    int rc = 0;
    if (n == 1) {  // PSD.jl:333-352
        for (int q = 0; q < nb; ++q) {
            PSD_LAUNCH(psd_scalar_product, psd_dim3(1), 64, 0, c->stream, (const double*)(dH + (size_t)q * p), p,
                       bws + 6 * sb + (size_t)q * (n + 8), bws + 7 * sb + (size_t)q * (n + 8));
        }
    } else {
        rc = iterate_dev(c, n, p, dH, dZ, wantT, wantZ, maxitfac, &st, s, mlog, nb, bws, pinfo.data());
        if (rc == 0) stats_from_state(s, st);
    }
    s->ms_iter = s->ms_total = t.stop(c->stream);
    if (rc == 0) {
        for (int q = 0; q < nb; ++q) {
            PSD_CHECK(psd_rt_d2h(wr + (size_t)q * n, bws + 6 * sb + (size_t)q * (n + 8), sizeof(double) * n, c->stream));
            PSD_CHECK(psd_rt_d2h(wi + (size_t)q * n, bws + 7 * sb + (size_t)q * (n + 8), sizeof(double) * n, c->stream));
            for (int j = 0; j < p; ++j) {
                PSD_CHECK(psd_rt_d2h(H[(size_t)q * p + j], dH + ((size_t)q * p + j) * nn, nn * 8, c->stream));
                if (wantZ) PSD_CHECK(psd_rt_d2h(Q[(size_t)q * p + j], dZ + ((size_t)q * p + j) * nn, nn * 8, c->stream));
            }
        }
        PSD_CHECK(psd_rt_sync(c->stream));
    }
    psd_rt_free(dH);
    if (dZ) psd_rt_free(dZ);
    psd_rt_free(bws);
    if (rc != 0) return *info = rc;
    int worst = 0;
    for (int q = 0; q < nb; ++q) {
        const int iq = (pinfo[q] == PSD_LIST_OVERFLOW) ? (PSD_INFO_RUNTIME + 77) : ((pinfo[q] != 0) ? (PSD_INFO_NOCONV + pinfo[q]) : 0);
        if (infos) infos[q] = iq;
        if (iq != 0 && worst == 0) worst = iq;
    }
    if (worst == 0 && st.info != 0)
        worst = (st.info == PSD_LIST_OVERFLOW) ? (PSD_INFO_RUNTIME + 77) : (PSD_INFO_NOCONV + st.info);
    return *info = worst;
}
}  // extern "C"
