#!/usr/bin/env python3
"""Headline benchmark: periodic Schur sweeps/s and algorithmic HBM GB/s on MI355X.

A *step* is one pass of the hot path over one synthetic problem: a complete `pschur!(A, :R; wantZ, wantT)`
(periodic Hessenberg-triangular reduction, Q formation, periodic QR iteration to full deflation) on operands that
are already resident in HBM when the timed region starts (`psd_d_pschur_dev`).  N=1 workload: the size
BASELINE.json quotes its target on, n=1024, p=64, Float64 (0.5 GiB of factors + 0.5 GiB of Schur vectors, past the
256 MiB Infinity Cache); `--n 512 --p 16` gives BASELINE configs[1] as a secondary line.  N>1: see DESIGN.md
"Multi-GPU"; value = all sweeps of all ranks / max-over-ranks time.

The accuracy gate (device checkpsd + eigenvalues against LAPACK on the explicit product) runs outside the timed
region; a failed gate makes the exit status non-zero.

After the headline, N = 1 runs BASELINE configs[1]..[4] once each (`configs` in the JSON line: time, sweeps or swaps,
algorithmic GB/s, fraction of the HBM peak, accuracy gate; `--no-configs` skips them) and the CPU baseline.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def period_slice(p, world, rank):
    """Contiguous slice [lo, hi) of the period owned by `rank` (SURVEY.md §8e partition; used by the sharded
    design, and by tests to pin the ownership rule)."""
    base, rem = divmod(p, world)
    lo = rank * base + min(rank, rem)
    return [lo, lo + base + (1 if rank < rem else 0)]


def aggregate(dist, seconds, units, device):
    """max-over-ranks time and sum-over-ranks units (the driver's contract)."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds, units
    if dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(u.item()))


def cpu_baseline(n, p, seed, eng, gpu_positions):
    """CPU baseline: the oracle (C++ restatement of the reference algorithm and loop order, g++ -O3, ONE thread — the
    reference is serial; test infrastructure) on rank 0, N = 1.
      (1) un-extrapolated: ONE COMPLETE pschur!(A,:R) of BASELINE configs[1] (n = 512, p = 16, Float64, wantT, wantZ: the
          reduction, Q, the whole iteration), about half a minute: `value` = its sweeps / its wall time;
      (2) a bounded sample of the headline size (what does not fit a bench run as a whole): the first K columns of the
          periodic Hessenberg reduction and S full-width double-shift sweeps on the Hessenberg-triangular form.
    An OpenMP leg over the rank-one updates of (2) existed in round 2 and was 4-6x SLOWER than one thread on this host
    (8 KiB columns: the fork/join per reflector costs more than the update); it is gone, not a baseline."""
    import ctypes as C

    import numpy as np
    import psdtest as pt

    lib = pt.oracle_lib()
    lib.psdo_d_phessenberg_cols.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    lib.psdo_set_sweep_cap.argtypes = [C.c_longlong]
    lib.psdo_set_threads.argtypes = [C.c_int]
    lib.psdo_set_threads(1)
    # (1) complete run at configs[1]
    n2, p2 = 512, 16
    A2 = pt.bench_factors(n2, p2, seed)
    t0 = time.time()
    po = pt.oracle_pschur(A2, "R")
    t_full = time.time() - t0
    sw2 = po.sweeplog[po.sweeplog[:, 0] == 0]
    nsw2 = int(len(sw2))
    pos2 = int((sw2[:, 2] - sw2[:, 1] + 1).sum()) if nsw2 else 0
    bytes2 = float(sum(2 * 8 * p2 * int(w) * (2 * n2 + 1) for w in (sw2[:, 2] - sw2[:, 1] + 1))) + \
        2 * 8 * p2 * (5.0 / 6.0) * n2 ** 3 + 2 * 8 * p2 * n2 ** 3 / 3.0
    oko, erro = pt.checkpsd(po, A2, thresh=100 * np.sqrt(n2 / 32))  # (the reference's own checker, restated in numpy)
    out = {"value": nsw2 / t_full, "unit": "sweeps/s", "cores": 1, "kind": "port",
           "sample": f"oracle (C++ restatement of the reference algorithm, g++ -O3, 1 thread): one COMPLETE pschur!(A,:R) "
                     f"n={n2} p={p2} Float64 wantT wantZ (BASELINE configs[1], the bench input at that size): {nsw2} sweeps, "
                     f"{pos2} chase positions in {t_full:.1f} s — not extrapolated",
           "complete_run": {"n": n2, "p": p2, "seconds": t_full, "sweeps": nsw2, "sweep_positions": pos2,
                            "sweeps_per_eigenvalue": nsw2 / n2, "algorithmic_GBps": bytes2 / t_full / 1e9,
                            "checkpsd_ok": bool(oko), "checkpsd_max_err_eps": float(np.max(erro))}}
    # (2) bounded sample at the headline size
    if (n, p) != (n2, p2):
        K, S = (6, 4) if n >= 768 else (max(8, n // 16), 12)
        As = pt.bench_factors(n, p, seed)
        A = pt.pack(As)
        tau = np.zeros((p, n))
        t0 = time.time()
        lib.psdo_d_phessenberg_cols(n, p, pt._dp(A), pt._dp(tau), K)
        t_h = time.time() - t0
        col_el = lambda i: (n - i + 1) * (n - i) + n * (n - i + 1)
        frac = sum(col_el(i) for i in range(1, K + 1)) / sum(col_el(i) for i in range(1, n))
        W = [a.copy(order="F") for a in As]
        Hs, _, _ = eng.phessenberg_(W)  # (the engine's own reduction supplies the Hessenberg-triangular form)
        H = pt.pack(Hs)
        Z = pt.pack([np.eye(n) for _ in range(p)])
        wr, wi = np.zeros(n), np.zeros(n)
        niter, nlog = C.c_int64(0), C.c_int64(0)
        log = np.zeros(3 * (S + 64), dtype=np.int32)
        lib.psdo_set_sweep_cap(S)
        t0 = time.time()
        rc = lib.psdo_d_pschur_hess(n, p, pt._dp(H), pt._dp(Z), 1, 1, 30, pt._dp(wr), pt._dp(wi), C.byref(niter),
                                    log.ctypes.data_as(C.POINTER(C.c_int32)), S + 64, C.byref(nlog))
        t_s = time.time() - t0
        lib.psdo_set_sweep_cap(-1)
        lg = log[: 3 * min(nlog.value, S + 64)].reshape(-1, 3)
        sw = lg[lg[:, 0] == 0]
        pos = int((sw[:, 2] - sw[:, 1] + 1).sum()) if len(sw) else 0
        pos_per_s = pos / t_s if t_s > 0 else 0.0
        out["headline_sample"] = {
            "n": n, "p": p, "hessenberg_columns": K, "hessenberg_seconds": t_h, "hessenberg_fraction_of_bytes": frac,
            "sweeps": int(len(sw)), "sweep_seconds": t_s, "sweep_width": int(sw[0, 2] - sw[0, 1] + 1) if len(sw) else 0,
            "sweeps_per_s": len(sw) / t_s if t_s > 0 else None, "sweep_positions_per_s": pos_per_s, "rc": int(rc),
            "estimated_time_to_solution_s": {"hessenberg": t_h / frac, "iteration_at_gpu_position_count": gpu_positions / pos_per_s if pos_per_s else None,
                                             "note": "extrapolated from the sample by algorithmic bytes / chase positions (the "
                                                     "iteration estimate charges the CPU with the GPU's position count)"}}
    return out


def cycle_shares(st):
    """In-kernel s_memtime accounting of the chase kernel (psd_stats.step_cycles): share of decide / window load /
    chase / window store in the kernel's own time, and the shader clock derived from s_memrealtime."""
    c = list(st.step_cycles)
    if not c[4] or not (c[0] + c[1] + c[2] + c[3]):
        return None
    tot = c[0] + c[1] + c[2] + c[3]  # (all workgroups of a tick: leaders and cursors)
    out = {"decide": c[0] / tot, "window_load": c[1] / tot, "chase": c[2] / tot, "window_store": c[3] / tot}
    if c[5]:
        out["shader_clock_GHz"] = c[4] / (c[5] * 10e-9) / 1e9  # s_memrealtime ticks at 100 MHz
    return out


def pmc_traffic(n, p):
    """HBM traffic of the iteration kernels from the committed PMC passes at THIS size (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs of the standalone driver, the guide's gfx950 correction applied): the whole record of the
    newest pass, with ITS OWN algorithmic bytes, tick and sweep counts next to the traffic, so that traffic / algorithmic
    is read from one run (counters cannot be collected from inside this process).  None if no pass exists."""
    for rnd in ("r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", rnd, f"pmc_traffic_{n}x{p}.json")
        if os.path.exists(path):
            with open(path) as fh:
                d = json.load(fh)
            d["file"] = f"profiles/{rnd}/pmc_traffic_{n}x{p}.json"
            return d
    return None


def reference_equivalent(n, p):
    """Sweep count and algorithmic sweep bytes of the REFERENCE's iteration (one shift pair per sweep, one active range
    at a time) on the bench input, measured once on the GPU with the multishift trains off (tools/ref_equiv.py; that
    mode tracks the CPU oracle sweep for sweep: tests/test_gpu_headline.py) and committed under profiles/."""
    for rnd in ("r04", "r03"):
        path = os.path.join(ROOT, "profiles", rnd, f"reference_equivalent_{n}x{p}.json")
        if os.path.exists(path):
            with open(path) as fh:
                d = json.load(fh)
            d["file"] = f"profiles/{rnd}/reference_equivalent_{n}x{p}.json"
            return d
    return None


def psd_copy(ps):
    """deep copy of a PeriodicSchur (ordschur_ works in place)"""
    import psd_amd

    return psd_amd.PeriodicSchur([t.copy(order="F") for t in ps.Ts], [z.copy(order="F") for z in ps.Z], ps.values.copy(),
                                 ps.orientation, ps.schurindex, stats=ps.stats, sweeplog=ps.sweeplog)


def run_configs(eng, torch, device):
    """BASELINE configs[1]..[4] once each through the host entry points (operands start in host memory, as a Julia
    caller's would: `ms` is the engine's own device time, `wall_s` includes the PCIe copies), each with its accuracy
    gate: device checkpsd at 100*sqrt(n/32) eps and eigenvalues against LAPACK on the explicitly formed product at
    1e-10 * ||prod||_2 (configs[3]: inverted factors make the explicit product's own eigenvalues uncertain by
    cond * eps, which is added to the tolerance)."""
    import numpy as np
    import psdtest as pt

    out = {}

    def prod_gpu(As, S=None):
        P = None
        for l, a in enumerate(As):
            f = torch.from_numpy(np.ascontiguousarray(a)).to(device)
            if S is not None and not S[l]:
                f = torch.linalg.inv(f)
            P = f if P is None else P @ f
        return P.cpu().numpy()

    def frac(bytes_, ms):
        g = bytes_ / (ms * 1e-3) / 1e9 if ms else None
        return g, (g / HBM_PEAK_GBS if g else None)

    # configs[1]: pschur!(A,:R) n = 512, p = 16, Float64
    n, p = 512, 16
    As = pt.bench_factors(n, p, 1234 + 2)
    t0 = time.time()
    ps = eng.pschur(As, "R")
    wall = time.time() - t0
    st = ps.stats
    ok, err = eng.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    P = prod_gpu(As)
    le = pt.match_eigs(np.linalg.eigvals(P), ps.values) / np.linalg.norm(P, 2)
    by = st.bytes_sweeps + st.bytes_hess + st.bytes_formq
    g, f = frac(by, st.ms_total - st.ms_copy)
    out["cfg2"] = {"config": "configs[1]: pschur!(A,:R) n=512 p=16 Float64", "ms": st.ms_total - st.ms_copy, "wall_s": wall,
                   "phase_ms": {"hessenberg": st.ms_hess, "formq": st.ms_formq, "iteration": st.ms_iter},
                   "sweeps": st.nsweeps, "sweeps_per_eigenvalue": st.nsweeps / n, "algorithmic_GBps": g, "frac": f,
                   "checkpsd_max_err_eps": float(err.max()), "eig_rel_err": float(le), "gate_ok": bool(ok and le <= 1e-10)}
    # configs[2]: pschur!(A,:R) n = 1024, p = 64, ComplexF64 (one GPU; the period shard of the same call: --gpus N --dtype c128)
    n, p = 1024, 64
    As = pt.bench_factors(n, p, 1234 + 3, dtype=np.complex128)
    t0 = time.time()
    ps = eng.pschur(As, "R")
    wall = time.time() - t0
    st = ps.stats
    ok, err = eng.checkpsd(ps, As, thresh=100 * np.sqrt(n / 32))
    P = prod_gpu(As)
    le = pt.match_eigs(np.linalg.eigvals(P), ps.values) / np.linalg.norm(P, 2)
    by = st.bytes_sweeps + st.bytes_hess + st.bytes_formq
    g, f = frac(by, st.ms_total - st.ms_copy)
    out["cfg3"] = {"config": "configs[2]: pschur!(A,:R) n=1024 p=64 ComplexF64, one GPU", "ms": st.ms_total - st.ms_copy, "wall_s": wall,
                   "phase_ms": {"hessenberg": st.ms_hess, "formq": st.ms_formq, "iteration": st.ms_iter},
                   "sweeps": st.nsweeps, "sweeps_per_eigenvalue": st.nsweeps / n, "algorithmic_GBps": g, "frac": f,
                   "checkpsd_max_err_eps": float(err.max()), "eig_rel_err": float(le), "gate_ok": bool(ok and le <= 1e-10)}
    del As, ps, P
    # configs[3]: pschur!(A,S,:R) n = 512, p = 32, Float64, alternating signature
    n, p = 512, 32
    S = [True] + [bool(q % 2 == 0) for q in range(1, p)]
    A = pt.bench_factors(n, p, seed=4)
    t0 = time.time()
    ps = eng.pschur_([a.copy(order="F") for a in A], "R", S=S)
    wall = time.time() - t0
    st = ps.stats
    ok, err = eng.checkpsd(ps, A, thresh=100 * np.sqrt(n / 32), S=S)
    P = prod_gpu(A, S)
    nP = np.linalg.norm(P, 2)
    tol = max(1e-10, 100 * np.linalg.cond(P) * pt.EPS)
    le = pt.match_eigs(np.linalg.eigvals(P), ps.values) / nP
    by = st.bytes_sweeps + st.bytes_hess
    g, f = frac(by, st.ms_total - st.ms_copy)
    out["cfg4"] = {"config": "configs[3]: pschur!(A,S,:R) n=512 p=32 Float64, S = [T,F,T,F,...]", "ms": st.ms_total - st.ms_copy,
                   "wall_s": wall, "phase_ms": {"hessenberg": st.ms_hess, "hessenberg_stage1": st.ms_formq, "iteration": st.ms_iter},
                   "sweeps": st.nsweeps, "sweeps_per_eigenvalue": st.nsweeps / n, "algorithmic_GBps": g, "frac": f,
                   "checkpsd_max_err_eps": float(err.max()), "eig_rel_err": float(le), "eig_tol": float(tol),
                   "gate_ok": bool(ok and le <= tol)}
    # configs[4]: pschur!(A,:L) n = 1024, p = 16, then ordschur! of the n/4 eigenvalues of largest modulus
    n, p = 1024, 16
    As = pt.bench_factors(n, p, 1234 + 5)
    t0 = time.time()
    ps = eng.pschur(As, "L")
    wall_ps = time.time() - t0
    s0 = ps.stats
    ms_ps, sweeps_ps = s0.ms_total - s0.ms_copy, s0.nsweeps
    lam0 = ps.values.copy()
    order = np.argsort(-np.abs(lam0), kind="stable")
    select = np.zeros(n, dtype=bool)
    select[order[: n // 4]] = True
    for i in np.where(lam0.imag != 0)[0]:  # conjugates closed (rordschur.jl:44-58)
        if select[i]:
            select[i + 1 if lam0[i].imag > 0 else i - 1] = True
    m = int(select.sum())
    # (a second copy of the decomposition for the selection that has to move something: see below)
    ps_b = psd_copy(ps)
    t0 = time.time()
    ps1 = eng.ordschur_(ps, select)
    wall_or = time.time() - t0
    s1 = ps1.stats
    ok, err = eng.checkpsd(ps1, As, thresh=100 * np.sqrt(n / 32))
    sc = abs(lam0).max()
    e_sel = pt.match_eigs(lam0[select], ps1.values[:m]) / sc
    e_rest = pt.match_eigs(lam0[~select], ps1.values[m:]) / sc
    swaps = s1.nsweeps
    by_or = swaps * 2 * 8 * p * 3 * n * 2.5  # (2 E p 3 n m per adjacent swap, m = 2..4)
    g, f = frac(by_or, s1.ms_total)
    out["cfg5"] = {"config": "configs[4]: pschur!(A,:L) n=1024 p=16 Float64, then ordschur!(P, select = n/4 largest |lambda|)",
                   "ms": ms_ps + s1.ms_total, "pschur_ms": ms_ps, "pschur_sweeps": sweeps_ps, "pschur_wall_s": wall_ps,
                   "ordschur_ms": s1.ms_total, "ordschur_wall_s": wall_or, "selected": m, "swaps": swaps,
                   "swaps_per_s": swaps / (s1.ms_total * 1e-3) if s1.ms_total else None, "algorithmic_GBps": g, "frac": f,
                   "checkpsd_max_err_eps": float(err.max()), "selected_match": float(e_sel), "rest_match": float(e_rest),
                   "gate_ok": bool(ok and e_sel <= 1e-10 and e_rest <= 1e-10)}
    # The same decomposition, select = the n/4 eigenvalues of SMALLEST modulus: pschur!(:L) delivers the spectrum nearly
    # sorted by decreasing modulus, so BASELINE's selection above needs a handful of swaps; this one carries every
    # selected block across three quarters of the matrix (the workload of rordschur.jl:99,141-251 / sylswap.jl).
    order = np.argsort(np.abs(lam0), kind="stable")
    select = np.zeros(n, dtype=bool)
    select[order[: n // 4]] = True
    for i in np.where(lam0.imag != 0)[0]:
        if select[i]:
            select[i + 1 if lam0[i].imag > 0 else i - 1] = True
    m = int(select.sum())
    t0 = time.time()
    ps2 = eng.ordschur_(ps_b, select)
    wall_or = time.time() - t0
    s2 = ps2.stats
    ok, err = eng.checkpsd(ps2, As, thresh=100 * np.sqrt(n / 32))
    e_sel = pt.match_eigs(lam0[select], ps2.values[:m]) / sc
    e_rest = pt.match_eigs(lam0[~select], ps2.values[m:]) / sc
    swaps = s2.nsweeps
    by_or = swaps * 2 * 8 * p * 3 * n * 2.5
    g, f = frac(by_or, s2.ms_total)
    out["cfg5_smallest"] = {"config": "the decomposition of configs[4], ordschur!(P, select = n/4 smallest |lambda|): every selected block "
                                      "crosses three quarters of the matrix",
                            "ms": s2.ms_total, "ordschur_ms": s2.ms_total, "ordschur_wall_s": wall_or, "selected": m, "swaps": swaps,
                            "swaps_per_s": swaps / (s2.ms_total * 1e-3) if s2.ms_total else None, "algorithmic_GBps": g, "frac": f,
                            "checkpsd_max_err_eps": float(err.max()), "selected_match": float(e_sel), "rest_match": float(e_rest),
                            "gate_ok": bool(ok and e_sel <= 1e-10 and e_rest <= 1e-10)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--order", dest="n", type=int, default=1024)
    ap.add_argument("--p", "--period", dest="p", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded",
                    help="N > 1: 'sharded' = ONE problem, Schur vectors split by period over the ranks (strong scaling); "
                         "'replicas' = one independent problem per rank (weak scaling)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--dtype", choices=["f64", "c128"], default="f64",
                    help="c128: BASELINE configs[2] (pschur! of ComplexF64 factors) as the timed workload, e.g. period-sharded: "
                         "--gpus N --dtype c128")
    ap.add_argument("--no-configs", action="store_true", help="skip the one-shot runs of BASELINE configs[1]..[4] behind the headline")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import psd_amd
    import psdtest as pt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # (one process per GPU under torch.distributed.run: a bare `python bench.py --gpus 8` would measure ONE GPU and
        #  label it 8)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    backend = os.environ.get("PSD_BENCH_BACKEND", "nccl")  # ("gloo": rehearsal of the N > 1 path on a one-GPU box)
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    n, p = args.n, args.p
    seed = 1234 + 2  # BASELINE config index 2
    eng = psd_amd.Engine(device=local_rank)
    sharded = world > 1 and args.mode == "sharded"
    if sharded:
        import importlib.util

        spec = importlib.util.spec_from_file_location(
            "psd_sharded", os.path.join(ROOT, "periodicschurdecompositions.jl_amd", "sharded.py"))
        shmod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(shmod)
        eng.set_shard(rank, world)
    cplx = args.dtype == "c128"
    if cplx:
        seed = 1234 + 3  # BASELINE config index 3 (configs[2])
    As = pt.bench_factors(n, p, seed, dtype=np.complex128) if cplx else pt.bench_factors(n, p, seed)
    host = torch.from_numpy(pt.pack(As, np.complex128 if cplx else np.float64))
    total = args.steps + args.warmup
    bufs = [host.to(device) for _ in range(total)]
    zbufs = [torch.zeros_like(bufs[0]) for _ in range(total)]
    torch.cuda.synchronize()

    def run(k):
        r = (eng.zpschur_dev if cplx else eng.pschur_dev)(bufs[k].data_ptr(), n, p, "R", dZ_ptr=zbufs[k].data_ptr())
        if sharded:  # the one collective of the sharded mode: every rank ends up with every Z_j (RCCL all-gather)
            shmod.allgather_z_device(dist, zbufs[k].view(p, n, n), p, world, rank)
        return r

    for k in range(args.warmup):
        run(k)
    eng.set_profile(not args.no_kernel_events)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    results = [run(args.warmup + k) for k in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    eng.set_profile(False)

    exit_code = 0
    sweeps = sum(st.nsweeps for (_, _, st, _) in results)
    elapsed_max, sweeps_all = aggregate(dist if world > 1 else None, elapsed, sweeps, device)
    if sharded:
        sweeps_all = sweeps  # ONE problem: every rank ran the same (replicated) chain, its sweeps count once
    # Beside the sharded line: the same K steps as N independent replicas (one problem per rank, no collective) — the
    # throughput mode of a path whose chains do not shard.  Reported as an extra object.
    replicas = None
    if sharded:
        eng.set_shard(0, 1)
        rb = [host.to(device) for _ in range(args.steps)]
        rz = [torch.zeros_like(rb[0]) for _ in range(args.steps)]
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        rres = [(eng.zpschur_dev if cplx else eng.pschur_dev)(rb[k].data_ptr(), n, p, "R", dZ_ptr=rz[k].data_ptr()) for k in range(args.steps)]
        torch.cuda.synchronize()
        dist.barrier()
        el_r = time.perf_counter() - t1
        sw_r = sum(st.nsweeps for (_, _, st, _) in rres)
        el_r_max, sw_r_all = aggregate(dist, el_r, sw_r, device)
        replicas = {"value": sw_r_all / el_r_max, "unit": "sweeps/s", "scaling": "weak", "ms_per_step": 1e3 * el_r_max / args.steps,
                    "parallelism": "replicas x%d (one independent problem per rank, no collective)" % world}
        del rb, rz
        eng.set_shard(rank, world)

    def device_gate(lam_):
        """checkpsd of the last timed step's result (device; ComplexF64 through the host entry, which stages the blocks)"""
        thresh_ = 100 * np.sqrt(n / 32)
        if not cplx:
            dA0_ = host.to(device)
            r_ = eng.checkpsd_dev(bufs[-1].data_ptr(), zbufs[-1].data_ptr(), dA0_.data_ptr(), n, p, "R", 1, thresh=thresh_)
            del dA0_
            return r_
        Ts_ = [np.asfortranarray(bufs[-1][j].cpu().numpy().T) for j in range(p)]
        Zs_ = [np.asfortranarray(zbufs[-1][j].cpu().numpy().T) for j in range(p)]
        ps_ = psd_amd.PeriodicSchur(Ts_, Zs_, lam_, "R", 1)
        ok_, err_, orth_, tri_ = eng.checkpsd(ps_, As, thresh=thresh_, details=True)
        return ok_, err_, orth_, tri_

    # In the sharded mode every rank checks the decomposition it holds (its own T, the gathered Z) against the input and
    # the verdicts are combined: a rank whose chains had taken another path than the others' would fail here.
    shard_gate_ok = True
    if sharded:
        okr, _, _, _ = device_gate(results[-1][0])
        flag = torch.tensor([1.0 if okr else 0.0], dtype=torch.float64, device=None if backend != "nccl" else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        shard_gate_ok = bool(flag.item() > 0.5)
    if rank == 0:
        st = results[-1][2]
        lam = results[-1][0]
        # accuracy gate on the last timed step (outside the timed region): eigenvalues against LAPACK on the explicit
        # product, and the reference's checkpsd (diagnostics.jl:190-263) evaluated on the device
        P = pt.product(As)
        lam_ref = np.linalg.eigvals(P)
        lam_err = pt.match_eigs(lam_ref, lam) / np.linalg.norm(P, 2)
        thresh = 100 * np.sqrt(n / 32)
        ok, err, orth, tri = device_gate(lam)
        gate_ok = bool(bool(ok) and float(lam_err) <= 1e-10 and shard_gate_ok)

        nwin = sum(s.nwindows for (_, _, s, _) in results)
        nlaunch = sum(s.nlaunch_step for (_, _, s, _) in results)
        ntrain = sum(s.reserved for (_, _, s, _) in results)
        bytes_sw = sum(s.bytes_sweeps for (_, _, s, _) in results)
        ms_iter = sum(s.ms_iter for (_, _, s, _) in results)
        ms_hess = sum(s.ms_hess for (_, _, s, _) in results)
        ms_formq = sum(s.ms_formq for (_, _, s, _) in results)
        ksamples = sum(s.step_kernel_samples for (_, _, s, _) in results)
        kms = (sum(s.step_kernel_ms_avg * s.step_kernel_samples for (_, _, s, _) in results) / ksamples) if ksamples else None
        # One tick of the iteration = one launch of the chase kernel (one window of every bulge in flight) + the bulk-update
        # launches that apply its transformations to H and Z.  The algorithmic bytes are those of the bulk update, so the
        # roofline figure divides by the duration of the WHOLE tick (HIP events around the iteration phase on the engine's
        # stream / ticks), not by the chase launch alone; the latter (round 1's definition) is kept beside it.
        bytes_per_launch = bytes_sw / max(nlaunch, 1)
        tick_ms = ms_iter / max(nlaunch, 1)
        roof = None
        pmc = pmc_traffic(n, p)
        refeq = reference_equivalent(n, p)
        if kms and tick_ms > 0:
            achieved = bytes_per_launch / (tick_ms * 1e-3) / 1e9
            step_only = bytes_per_launch / (kms * 1e-3) / 1e9
            hess_gbs = st.bytes_hess / (st.ms_hess * 1e-3) / 1e9 if st.ms_hess else None
            links = (n - 1) * p
            roof = {"bound": "hbm", "kernel": "psd_rq_step_mb + psd_rq_apply_wl / _wl2 (one tick: chase launch + the bulk updates on its critical path)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": (pmc or {}).get("traffic_bytes_per_step_launch"),
                    "traffic_pass": pmc,
                    "alg_bytes_per_launch": bytes_per_launch, "avg_tick_ms": tick_ms, "ticks": nlaunch,
                    "windows_per_launch": nwin / max(nlaunch, 1),
                    "chase_launch_only": {"avg_launch_ms": kms, "launch_samples": ksamples, "achieved": step_only,
                                          "frac": step_only / HBM_PEAK_GBS,
                                          "note": "round 1's definition: the tick's algorithmic bytes / the chase launch alone"},
                    "hessenberg_link": {"kernel": "psd_hess2_link", "achieved": hess_gbs,
                                        "frac": hess_gbs / HBM_PEAK_GBS if hess_gbs else None,
                                        "alg_bytes_per_launch": st.bytes_hess / links if links else None,
                                        "avg_launch_ms": st.ms_hess / links if links else None,
                                        "traffic": None,
                                        "traffic_note": "no counter evidence: a counter pass over a process that runs the multi-stream reduction ends in the "
                                                        "profiler (profiles/r02/pmc_sigsegv_analysis.md, round-4 amendment); the one-stream form measured "
                                                        "1.60 x algorithmic per link at p = 8 (profiles/r02/pmc_traffic_hess2_1024x8.json)",
                                        "note": "one chain launch per link, consecutive launches overlapping on two streams (DESIGN section 0e; the panel "
                                                "updates of 24 links per launch run on a third, CU-masked stream beside them); algorithmic bytes "
                                                "16*(m*(m+1) + n*m) per link / HIP-event duration of the whole reduction per link"},
                    "reference_equivalent": None if refeq is None else {
                        "sweeps": refeq["sweeps"], "bytes": refeq["bytes_sweeps"], "sweeps_per_eigenvalue": refeq["sweeps"] / n,
                        "achieved": refeq["bytes_sweeps"] / (ms_iter / args.steps * 1e-3) / 1e9,
                        "frac": refeq["bytes_sweeps"] / (ms_iter / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "whole_call_frac": (refeq["bytes_sweeps"] + st.bytes_hess + st.bytes_formq) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                        "file": refeq["file"],
                        "note": "the sweep bytes of the REFERENCE's iteration on this input (one shift pair per sweep; measured "
                                "once with the trains off) / the benched iteration time: does not grow when the default mode "
                                "spends more sweeps per eigenvalue"},
                    "note": "algorithmic bytes of the sweep windows one tick chases (one window of every bulge in flight: "
                            "the cursors of the multishift trains of all active ranges; 2*8*p*w*(2n+1) per sweep) / "
                            "HIP-event duration of the tick; every bulge is latency-bound on its serial reflector chain, "
                            "the chase advances all p factors of a position at once (scan chase, DESIGN section 0), the bulk updates run at 2-4.5 TB/s"}
        out = {
            "metric": "PSD sweeps/sec (pschur! n=%d p=%d %s, Hessenberg+Q+iteration, operands in HBM)" % (n, p, "ComplexF64" if cplx else "Float64"),
            "value": sweeps_all / elapsed_max,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "c128" if cplx else "f64",
            "data": "synthetic",
            "config": {"workload": "%spschur!(A,:R) N=%d p=%d %s wantT wantZ, A_j = I + 0.5*G_j/sqrt(n)"
                                   % (("configs[2]: " if (n, p) == (1024, 64) else "") if cplx else
                                      ("north_star target size: " if (n, p) == (1024, 64) else
                                       ("configs[1]: " if (n, p) == (512, 16) else "")), n, p, "ComplexF64" if cplx else "Float64"),
                       "seed": seed, "parallelism": ("1 GPU" if world == 1 else
                                       ("period-sharded x%d (Z_j by slices of the period, chains replicated, one "
                                        "all-gather of Z)" % world if sharded else "replicas x%d" % world)), "window": st.window},
            "sweeps_per_step": sweeps / args.steps,
            "sweeps_per_eigenvalue": sweeps / args.steps / n,
            "sweeps_in_multishift_trains_per_step": ntrain / args.steps,
            "chase_kernel_cycle_shares": cycle_shares(st),
            "phase_ms_per_step": {"hessenberg": ms_hess / args.steps, "formq": ms_formq / args.steps,
                                  "iteration": ms_iter / args.steps},
            "algorithmic_GBps": {"sweeps": bytes_sw / (ms_iter * 1e-3) / 1e9 if ms_iter else None,
                                 "hessenberg": st.bytes_hess / (st.ms_hess * 1e-3) / 1e9 if st.ms_hess else None,
                                 "whole_call": (bytes_sw / args.steps + st.bytes_hess + st.bytes_formq) * args.steps / elapsed / 1e9},
            "time_to_solution_s": elapsed_max / args.steps,
            "accuracy": {"gate_ok": gate_ok, "eig_rel_err_vs_numpy_prod": float(lam_err), "eig_tol": 1e-10, "checkpsd_ok": bool(ok),
                         "checkpsd_max_err_eps": float(err.max()), "checkpsd_thresh_eps": thresh,
                         "orth_max_over_eps_n": float(orth.max() / (pt.EPS * n)), "evaluated": "device (psd_d_checkpsd_dev)"},
            "roofline": roof,
        }
        if replicas is not None:
            out["replicas"] = replicas
        if world == 1 and not args.no_configs and not cplx:
            del bufs, zbufs
            torch.cuda.empty_cache()
            out["configs"] = run_configs(eng, torch, device)
            if not all(c["gate_ok"] for c in out["configs"].values()):
                gate_ok = False
        if world == 1 and not args.no_cpu_baseline and not cplx:
            positions = sum(int((lg[lg[:, 0] == 0][:, 2] - lg[lg[:, 0] == 0][:, 1] + 1).sum()) for (_, _, _, lg) in results)
            out["sweep_positions_per_step"] = positions / args.steps
            out["cpu_baseline"] = cpu_baseline(n, p, seed, eng, positions / args.steps)
            # the device's checkpsd residual at configs[1] over the CPU oracle's on the same input
            if "configs" in out and "cfg2" in out["configs"]:
                eo = out["cpu_baseline"]["complete_run"]["checkpsd_max_err_eps"]
                out["configs"]["cfg2"]["checkpsd_max_err_eps_oracle"] = eo
                out["configs"]["cfg2"]["residual_over_oracle"] = out["configs"]["cfg2"]["checkpsd_max_err_eps"] / eo if eo else None
        print(json.dumps(out))
        if not gate_ok:
            sys.stderr.write("bench.py: ACCURACY GATE FAILED\n")
            exit_code = 1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
