#!/usr/bin/env python3
"""Builds tools/micro/c2_bench (a standalone timing harness of the two-wave chase loop, psd_c2_run of psd_real_qr.h) from
the product source and runs it: shader cycles per chain link for the full loop and for stripped variants.
usage: python tools/micro/c2_bench.py [p W]      (on the GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "periodicschurdecompositions.jl_amd", "csrc", "psd_real_qr.h")).read()
a = src.index("struct psd_c2 {\n    int cmd;")
b = src.index("// wavefront A's side of a run: publish it")
code = src[a:b]
hdr = '''#include "%s/periodicschurdecompositions.jl_amd/csrc/psd_scalar.h"
#include <vector>
#include <cstdio>
enum { PSD_TR_R3 = 3, PSD_TR_H2 = 2, PSD_TR_R2 = 4, PSD_TR_G = 5 };
#define PSD_TR_CAP 64
#define PSD_STEP_NT 64
struct psd_tr { int pos; int kind; double c0, c1, c2; };
PSD_D void psd_tr_store_global(psd_tr* dst, const psd_tr& tr) {
    __attribute__((address_space(1))) double* q = (__attribute__((address_space(1))) double*)dst;
    q[0] = __hiloint2double(tr.kind, tr.pos); q[1] = tr.c0; q[2] = tr.c1; q[3] = tr.c2;
}
#ifdef STAMPS
__device__ long long g_st[8], g_last;
#define PSD_C2_STAMP(i) do { const long long now__ = __builtin_amdgcn_s_memtime(); if (threadIdx.y == 0) { g_st[i] += now__ - g_last; g_last = now__; } } while (0)
#endif
#ifdef NOBARRIER
#undef PSD_PAIR_BARRIER
#define PSD_PAIR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
''' % ROOT
tail = r'''
__global__ void __launch_bounds__(128) kern(const psd_c2* Cp, const double* init, int ndoubles, int reps, int roles_mask, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char psd_lds[];
    double* wb = (double*)psd_lds;
    for (int q = threadIdx.x + 64 * threadIdx.y; q < ndoubles; q += 128) wb[q] = init[q];
    __syncthreads();
    psd_c2 C = *Cp;
    C.tr = C.tr + (size_t)blockIdx.x * C.p * PSD_TR_CAP;
    const int role = (threadIdx.y == 0) ? 1 : 2;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (role & roles_mask) psd_c2_run(C, role);
        else for (int s = 0; s < C.npos * C.p + PSD_C2_LAG + 1; ++s) PSD_PAIR_BARRIER();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x * 2 + threadIdx.y] = t1 - t0;
}
int main(int argc, char** argv) {
    const int p = argc > 1 ? atoi(argv[1]) : 64, W = argc > 2 ? atoi(argv[2]) : 17;
    const int ld = ((W + 1) & 1) ? W + 2 : W + 1, bsz = W * ld, npos = W - 4, nblk = 64, reps = 20;
    psd_c2 C;
    C.cmd = 1; C.ld = ld; C.bsz = bsz; C.bs = 100; C.be = 100 + W - 1; C.p = p; C.l = 1; C.i = 1000; C.ks = 101; C.npos = npos;
    C.c1max = C.be; C.r0 = C.bs; C.n1 = 0; C.nj = 0; C.wboff = 0; C.v0 = 0.3; C.v1 = 0.2; C.v2 = 0.1;
    const int nd = p * bsz;
    std::vector<double> h(nd);
    for (int j = 0; j < p; ++j)
        for (int c = 0; c < W; ++c)
            for (int r = 0; r < ld; ++r) {
                double v = 0.01 * (((r * 7 + c * 13 + j * 3) % 17) - 8);
                if (r == c) v += 1.0;
                if (r > c + (j == 0 ? 1 : 0)) v = 0.0;
                h[(size_t)j * bsz + c * ld + r] = v;
            }
    double* dinit; psd_c2* dC; psd_tr* dtr; long long* dcyc;
    hipMalloc(&dinit, nd * 8); hipMalloc(&dC, sizeof(C)); hipMalloc(&dtr, sizeof(psd_tr) * (size_t)nblk * p * PSD_TR_CAP); hipMalloc(&dcyc, 16 * nblk);
    C.tr = dtr;
    hipMemcpy(dinit, h.data(), nd * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, &C, sizeof(C), hipMemcpyHostToDevice);
    const size_t lds = (size_t)nd * 8 + 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const char* names[3] = {"A+B", "A only (B at the barriers)", "B only"};
    const int masks[3] = {3, 1, 2};
    for (int v = 0; v < 3; ++v) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(64, 2), lds, 0, dC, dinit, nd, reps, masks[v], dcyc);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(64, 2), lds, 0, dC, dinit, nd, reps, masks[v], dcyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc[2 * 64]; hipMemcpy(hc, dcyc, 16 * nblk, hipMemcpyDeviceToHost);
        const double links = (double)reps * npos * p;
        printf("%-28s p=%d W=%d: %.1f us per window (%d positions), %.0f cycles per link (wave A), %.0f (wave B)  [%s]\n", names[v], p, W,
               1e3 * ms / reps, npos, hc[0] / links, hc[1] / links, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
'''
os.makedirs("/tmp/c2b", exist_ok=True)
for variant, flags in (("barrier", []), ("nobarrier", ["-DNOBARRIER"])):
    path = "/tmp/c2b/c2_%s.hip" % variant
    open(path, "w").write(hdr + code + tail)
    exe = "/tmp/c2b/c2_%s" % variant
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast"] + flags + [path, "-o", exe])
    print("==", variant, flush=True)
    if os.environ.get("C2_COMPILE_ONLY") != "1":
        subprocess.call([exe] + sys.argv[1:3])
