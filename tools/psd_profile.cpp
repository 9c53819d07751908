// Standalone driver for profiler runs (rocprofv3 --pmc crashes under the python/torch harness on this image).
// Usage: psd_profile [n] [p] [repeat] [d|z]  — solves pschur!(A,:R) on A_j = I + 0.5 G_j/sqrt(n) through the C ABI
// (z: ComplexF64, re and im of G_j ~ N(0, 1/2)).
//        psd_profile n p 1 H <file>  — reduces the same Float64 input to periodic Hessenberg-triangular form (phessenberg!)
//                                      and writes the factors to <file>;
//        psd_profile n p rep I <file> — reads them and runs pschur!(H1, Hs) alone: a process whose only kernels are the
//                                      iteration's (counter passes: the profiler's packet interception does not survive
//                                      the multi-stream Hessenberg reduction, profiles/r02/pmc_sigsegv_analysis.md).
// Linked DIRECTLY against libpsd_mi355x.so (round 1 dlopen()ed it after the profiler had initialised the GPU; the
// unfiltered --pmc run of that build died with a SIGSEGV on a profiler thread, gpurun_out/pmc_fetch.log).  With
// PSD_PROFILE_MAPS=<file> the process writes /proc/self/maps there once the library and the HIP runtime are up, so
// that the frames of a crash can be attributed to a library.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/psd_mi355x.h"

static uint64_t sm64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double gauss(uint64_t& s) {
    const double u1 = ((sm64(s) >> 11) + 0.5) / 9007199254740992.0, u2 = ((sm64(s) >> 11) + 0.5) / 9007199254740992.0;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512, p = argc > 2 ? atoi(argv[2]) : 16, rep = argc > 3 ? atoi(argv[3]) : 1;
    auto create = &psd_create;
    auto destroy = &psd_destroy;
    auto pschur = &psd_d_pschur;
    psd_ctx* ctx = nullptr;
    if (create(&ctx, 0) != 0) { fprintf(stderr, "psd_create failed\n"); return 3; }
    if (const char* mp = getenv("PSD_PROFILE_MAPS")) {
        FILE* in = fopen("/proc/self/maps", "r");
        FILE* out = fopen(mp, "w");
        if (in && out) {
            char line[1024];
            while (fgets(line, sizeof(line), in)) fputs(line, out);
        }
        if (in) fclose(in);
        if (out) fclose(out);
    }
    const size_t nn = (size_t)n * n;
    if (argc > 4 && argv[4][0] == 'z') {
        std::vector<std::vector<double>> A0(p, std::vector<double>(2 * nn)), A(p), Z(p, std::vector<double>(2 * nn));
        uint64_t s = 1237;
        for (int j = 0; j < p; ++j)
            for (size_t q = 0; q < nn; ++q) {
                A0[j][2 * q] = 0.5 * gauss(s) / std::sqrt(2.0 * n) + ((q % (n + 1)) == 0 ? 1.0 : 0.0);
                A0[j][2 * q + 1] = 0.5 * gauss(s) / std::sqrt(2.0 * n);
            }
        std::vector<double> alpha(2 * n), beta(n);
        std::vector<int32_t> asc(n);
        for (int r = 0; r < rep; ++r) {
            std::vector<double*> Ap(p), Zp(p);
            for (int j = 0; j < p; ++j) { A[j] = A0[j]; Ap[j] = A[j].data(); Zp[j] = Z[j].data(); }
            psd_stats st; int si = 0, info = 0;
            psd_z_pschur(ctx, n, p, Ap.data(), nullptr, 'R', 1, 1, 30, Zp.data(), alpha.data(), beta.data(), asc.data(), &si, &st, nullptr, 0, &info);
            printf("{\"n\": %d, \"p\": %d, \"dtype\": \"c128\", \"info\": %d, \"sweeps\": %d, \"windows\": %d, \"launches\": %d, \"ms_total\": %.3f, "
                   "\"ms_iter\": %.3f, \"ms_hess\": %.3f, \"bytes_sweeps\": %.0f}\n",
                   n, p, info, st.nsweeps, st.nwindows, st.nlaunch_step, st.ms_total, st.ms_iter, st.ms_hess, st.bytes_sweeps);
        }
        destroy(ctx);
        return 0;
    }
    std::vector<std::vector<double>> A0(p, std::vector<double>(nn)), A(p), Z(p, std::vector<double>(nn));
    std::vector<double> wr(n), wi(n);
    if (argc > 5 && argv[4][0] == 'I') {  // iteration alone, from a dumped Hessenberg-triangular form
        FILE* fh = fopen(argv[5], "rb");
        if (!fh) { fprintf(stderr, "cannot read %s\n", argv[5]); return 4; }
        for (int j = 0; j < p; ++j)
            if (fread(A0[j].data(), sizeof(double), nn, fh) != nn) { fprintf(stderr, "short file\n"); return 4; }
        fclose(fh);
        for (int r = 0; r < rep; ++r) {
            std::vector<double*> Hp(p), Zp(p);
            for (int j = 0; j < p; ++j) {
                A[j] = A0[j];
                Hp[j] = A[j].data();
                std::fill(Z[j].begin(), Z[j].end(), 0.0);
                for (int q = 0; q < n; ++q) Z[j][(size_t)q * n + q] = 1.0;
                Zp[j] = Z[j].data();
            }
            psd_stats st; int info = 0;
            psd_d_pschur_hess(ctx, n, p, Hp.data(), Zp.data(), 1, 1, 30, wr.data(), wi.data(), &st, nullptr, 0, &info);
            printf("{\"n\": %d, \"p\": %d, \"mode\": \"iteration alone\", \"info\": %d, \"sweeps\": %d, \"windows\": %d, \"launches\": %d, "
                   "\"ms_iter\": %.3f, \"bytes_sweeps\": %.0f}\n", n, p, info, st.nsweeps, st.nwindows, st.nlaunch_step, st.ms_iter, st.bytes_sweeps);
        }
        destroy(ctx);
        return 0;
    }
    uint64_t s = 1236;
    for (int j = 0; j < p; ++j)
        for (size_t q = 0; q < nn; ++q) A0[j][q] = 0.5 * gauss(s) / std::sqrt((double)n) + ((q % (n + 1)) == 0 ? 1.0 : 0.0);
    if (argc > 5 && argv[4][0] == 'H') {  // the Hessenberg-triangular form of the input, to a file
        std::vector<double*> Ap(p);
        std::vector<double> tau((size_t)p * n);
        for (int j = 0; j < p; ++j) { A[j] = A0[j]; Ap[j] = A[j].data(); }
        psd_stats st; int info = 0;
        psd_d_phessenberg(ctx, n, p, Ap.data(), tau.data(), &st, &info);
        FILE* fh = fopen(argv[5], "wb");
        if (!fh) { fprintf(stderr, "cannot write %s\n", argv[5]); return 4; }
        for (int j = 0; j < p; ++j) {
            for (int c = 0; c < n; ++c)
                for (int r = c + ((j == 0) ? 2 : 1); r < n; ++r) A[j][(size_t)c * n + r] = 0.0;  // (the reflectors below)
            fwrite(A[j].data(), sizeof(double), nn, fh);
        }
        fclose(fh);
        printf("{\"n\": %d, \"p\": %d, \"mode\": \"hessenberg form written\", \"info\": %d, \"ms_hess\": %.3f}\n", n, p, info, st.ms_hess);
        destroy(ctx);
        return 0;
    }
    for (int r = 0; r < rep; ++r) {
        std::vector<double*> Ap(p), Zp(p);
        for (int j = 0; j < p; ++j) { A[j] = A0[j]; Ap[j] = A[j].data(); Zp[j] = Z[j].data(); }
        psd_stats st; int si = 0, info = 0;
        pschur(ctx, n, p, Ap.data(), nullptr, 'R', 1, 1, 30, Zp.data(), wr.data(), wi.data(), &si, &st, nullptr, 0, &info);
        printf("{\"n\": %d, \"p\": %d, \"info\": %d, \"sweeps\": %d, \"windows\": %d, \"launches\": %d, \"ms_total\": %.3f, "
               "\"ms_iter\": %.3f, \"ms_hess\": %.3f, \"bytes_sweeps\": %.0f, \"bytes_hess\": %.0f, \"bytes_formq\": %.0f}\n",
               n, p, info, st.nsweeps, st.nwindows, st.nlaunch_step, st.ms_total, st.ms_iter, st.ms_hess, st.bytes_sweeps,
               st.bytes_hess, st.bytes_formq);
    }
    destroy(ctx);
    return 0;
}
