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


def cpu_baseline(n, p, seed, eng, gpu_positions, budget_cols=None, budget_sweeps=None):
    """CPU baseline on a BOUNDED sample of the same workload: the oracle (C++ restatement of the reference algorithm,
    g++ -O3, test infrastructure) runs (a) the first K columns of the periodic Hessenberg reduction (K * p reflector
    links on the full-size factors) and (b) S full-width double-shift sweeps of the periodic QR iteration (wantT,
    wantZ) on the Hessenberg-triangular form of the same input, once on one thread and — the Hessenberg sample, whose
    large rank-one updates are what a threaded BLAS would thread in the reference (householder.jl:215,252) — once on
    all host cores (OpenMP).  The 3-row updates of a sweep stay serial, as in the reference.  Rank 0, N = 1 only."""
    import ctypes as C

    import numpy as np
    import psdtest as pt

    lib = pt.oracle_lib()
    lib.psdo_d_phessenberg_cols.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    lib.psdo_set_sweep_cap.argtypes = [C.c_longlong]
    lib.psdo_set_threads.argtypes = [C.c_int]
    big = n >= 768
    K = budget_cols or (6 if big else max(8, n // 16))
    S = budget_sweeps or (4 if big else 12)
    As = pt.bench_factors(n, p, seed)
    # threads the process may actually use (the GPU box gives one GPU's share of the host's cores to a job)
    ncores = min(len(os.sched_getaffinity(0)), 64) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    ncores = int(os.environ.get('PSD_BENCH_THREADS', ncores))

    def hess_sample(threads):
        lib.psdo_set_threads(threads)
        A = pt.pack(As)
        tau = np.zeros((p, n))
        t0 = time.time()
        lib.psdo_d_phessenberg_cols(n, p, pt._dp(A), pt._dp(tau), K)
        dt = time.time() - t0
        lib.psdo_set_threads(1)
        return dt

    t_h1 = hess_sample(1)
    t_hN = hess_sample(ncores)
    # algorithmic bytes of columns 1..K against the whole reduction: sum_i [(n-i+1)(n-i) + n(n-i+1)] elements per factor
    col_el = lambda i: (n - i + 1) * (n - i) + n * (n - i + 1)
    frac = sum(col_el(i) for i in range(1, K + 1)) / sum(col_el(i) for i in range(1, n))
    # (b) sweeps on the Hessenberg-triangular form (the engine's own reduction of the same input supplies it)
    W = [a.copy(order="F") for a in As]
    Hs, _, _ = eng.phessenberg_(W)
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
    nsw = len(sw)
    pos_per_s = pos / t_s if t_s > 0 else 0.0
    est_hess = t_h1 / frac
    est_iter = gpu_positions / pos_per_s if pos_per_s else None
    return {"value": nsw / t_s, "unit": "sweeps/s", "cores": 1, "kind": "port",
            "sample": f"oracle (C++ restatement of the reference algorithm, g++ -O3 -fopenmp): {nsw} double-shift sweeps "
                      f"of width {int(sw[0, 2] - sw[0, 1] + 1) if nsw else 0} of pschur!(H1,Hs) n={n} p={p} Float64 wantT "
                      f"wantZ in {t_s:.1f} s (stopped by a sweep cap, rc={rc}), and the first {K} of {n - 1} columns of "
                      f"phessenberg! ({K * p} links, {100 * frac:.2f} % of the reduction's algorithmic bytes) in "
                      f"{t_h1:.1f} s on 1 thread / {t_hN:.1f} s on {ncores} threads; the job may use {ncores} of the host's {os.cpu_count()} cores",
            "sweep_positions_per_s": pos_per_s,
            "hessenberg_sample_s": {"threads_1": t_h1, f"threads_{ncores}": t_hN, "columns": K},
            "allcore": {"cores": ncores, "hessenberg_speedup": t_h1 / t_hN if t_hN > 0 else None,
                        "note": "sweeps are 3-row updates and stay serial, as in the reference"},
            "estimated_time_to_solution_s": {"hessenberg_1thread": est_hess, "hessenberg_allcore": t_hN / frac,
                                             "iteration_at_gpu_position_count": est_iter,
                                             "note": "extrapolated from the sample by algorithmic bytes / chase "
                                                     "positions; the iteration estimate charges the CPU with the "
                                                     "GPU's position count (multishift trains take more sweeps)"}}


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
    """HBM bytes per chase launch (step + apply kernels) from the committed PMC passes at THIS size (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs of this command, gfx950 correction of the guide applied;
    profiles/r02/pmc_traffic_<n>x<p>.json records the command, the sweep / tick counts of the counted run and the raw
    counters).  Counters cannot be collected from inside this process; None if no pass exists for the size."""
    root = os.path.dirname(os.path.abspath(__file__))
    for rnd in ("r02",):
        path = os.path.join(root, "profiles", rnd, f"pmc_traffic_{n}x{p}.json")
        if os.path.exists(path):
            with open(path) as fh:
                d = json.load(fh)
            return d.get("traffic_bytes_per_step_launch"), {k: d.get(k) for k in ("source", "sweeps", "ticks", "command")}
    return None, None


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
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import psd_amd
    import psdtest as pt

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    As = pt.bench_factors(n, p, seed)
    host = torch.from_numpy(pt.pack(As))
    total = args.steps + args.warmup
    bufs = [host.to(device) for _ in range(total)]
    zbufs = [torch.zeros_like(bufs[0]) for _ in range(total)]
    torch.cuda.synchronize()

    def run(k):
        r = eng.pschur_dev(bufs[k].data_ptr(), n, p, "R", dZ_ptr=zbufs[k].data_ptr())
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
        rres = [eng.pschur_dev(rb[k].data_ptr(), n, p, "R", dZ_ptr=rz[k].data_ptr()) for k in range(args.steps)]
        torch.cuda.synchronize()
        dist.barrier()
        el_r = time.perf_counter() - t1
        sw_r = sum(st.nsweeps for (_, _, st, _) in rres)
        el_r_max, sw_r_all = aggregate(dist, el_r, sw_r, device)
        replicas = {"value": sw_r_all / el_r_max, "unit": "sweeps/s", "scaling": "weak", "ms_per_step": 1e3 * el_r_max / args.steps,
                    "parallelism": "replicas x%d (one independent problem per rank, no collective)" % world}
        del rb, rz
        eng.set_shard(rank, world)

    if rank == 0:
        st = results[-1][2]
        lam = results[-1][0]
        # accuracy gate on the last timed step (outside the timed region): eigenvalues against LAPACK on the explicit
        # product, and the reference's checkpsd (diagnostics.jl:190-263) evaluated on the device
        P = pt.product(As)
        lam_ref = np.linalg.eigvals(P)
        lam_err = pt.match_eigs(lam_ref, lam) / np.linalg.norm(P, 2)
        dA0 = host.to(device)
        thresh = 100 * np.sqrt(n / 32)
        ok, err, orth, tri = eng.checkpsd_dev(bufs[-1].data_ptr(), zbufs[-1].data_ptr(), dA0.data_ptr(), n, p, "R", 1,
                                              thresh=thresh)
        del dA0
        gate_ok = bool(bool(ok) and float(lam_err) <= 1e-10)

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
        if kms and tick_ms > 0:
            achieved = bytes_per_launch / (tick_ms * 1e-3) / 1e9
            step_only = bytes_per_launch / (kms * 1e-3) / 1e9
            hess_gbs = st.bytes_hess / (st.ms_hess * 1e-3) / 1e9 if st.ms_hess else None
            links = (n - 1) * p
            roof = {"bound": "hbm", "kernel": "psd_rq_step_mb + psd_rq_apply_wl (one tick: chase launch + its bulk updates)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n, p)[0], "traffic_source": pmc_traffic(n, p)[1],
                    "alg_bytes_per_launch": bytes_per_launch, "avg_tick_ms": tick_ms, "ticks": nlaunch,
                    "windows_per_launch": nwin / max(nlaunch, 1),
                    "chase_launch_only": {"avg_launch_ms": kms, "launch_samples": ksamples, "achieved": step_only,
                                          "frac": step_only / HBM_PEAK_GBS,
                                          "note": "round 1's definition: the tick's algorithmic bytes / the chase launch alone"},
                    "hessenberg_link": {"kernel": "psd_hess2_link", "achieved": hess_gbs,
                                        "frac": hess_gbs / HBM_PEAK_GBS if hess_gbs else None,
                                        "alg_bytes_per_launch": st.bytes_hess / links if links else None,
                                        "avg_launch_ms": st.ms_hess / links if links else None,
                                        "note": "one chain launch per link on the main stream (the panel updates of 16 links per launch run on a "
                                                "second, CU-masked stream beside it); algorithmic bytes 16*(m*(m+1) + n*m) per link / "
                                                "HIP-event duration of the reduction per link"},
                    "note": "algorithmic bytes of the sweep windows one tick chases (one window of every bulge in flight: "
                            "the cursors of the multishift trains of all active ranges; 2*8*p*w*(2n+1) per sweep) / "
                            "HIP-event duration of the tick; every bulge is latency-bound on its serial reflector chain, "
                            "the bulk updates run at 2-3 TB/s (profiles/r02)"}
        out = {
            "metric": "PSD sweeps/sec (pschur! n=%d p=%d Float64, Hessenberg+Q+iteration, operands in HBM)" % (n, p),
            "value": sweeps_all / elapsed_max,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%spschur!(A,:R) N=%d p=%d Float64 wantT wantZ, A_j = I + 0.5*G_j/sqrt(n)"
                                   % ("north_star target size: " if (n, p) == (1024, 64) else
                                      ("configs[1]: " if (n, p) == (512, 16) else ""), n, p),
                       "seed": seed, "parallelism": ("1 GPU" if world == 1 else
                                       ("period-sharded x%d (Z_j by slices of the period, chains replicated, one "
                                        "all-gather of Z)" % world if sharded else "replicas x%d" % world)), "window": st.window},
            "sweeps_per_step": sweeps / args.steps,
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
        if world == 1 and not args.no_cpu_baseline:
            positions = sum(int((lg[lg[:, 0] == 0][:, 2] - lg[lg[:, 0] == 0][:, 1] + 1).sum()) for (_, _, _, lg) in results)
            out["sweep_positions_per_step"] = positions / args.steps
            out["cpu_baseline"] = cpu_baseline(n, p, seed, eng, positions / args.steps)
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
