#!/usr/bin/env python3
"""Headline benchmark: periodic Schur sweeps/s and algorithmic HBM GB/s on MI355X.

A *step* is one pass of the hot path over one synthetic problem: a complete `pschur!(A, :R; wantZ, wantT)`
(periodic Hessenberg-triangular reduction, Q formation, periodic QR iteration to full deflation) on operands that
are already resident in HBM when the timed region starts (`psd_d_pschur_dev`).  N=1 workload: BASELINE.json
configs[1], n=512, p=16, Float64.  N>1: one process per GPU, each solves an independent replica of the same
workload ("replicas only", DESIGN.md "Multi-GPU"); value = all sweeps of all ranks / max-over-ranks time.

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
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(u.item()))


def cpu_baseline(n, p, seed):
    """The oracle (C++ restatement of the reference algorithm, one thread) on the same workload, rank 0 only."""
    import psdtest as pt

    As = pt.bench_factors(n, p, seed)
    t0 = time.time()
    po = pt.oracle_pschur(As, "R")
    dt = time.time() - t0
    nsw = int((po.sweeplog[:, 0] == 0).sum())
    return {"value": nsw / dt, "unit": "sweeps/s", "cores": 1, "kind": "port",
            "sample": f"1 full pschur!(A,:R) n={n} p={p} Float64 wantZ wantT ({nsw} sweeps, {dt:.1f} s), "
                      f"C++ restatement of the reference algorithm, g++ -O3, 1 thread of {os.cpu_count()} host cores",
            "seconds": dt, "phase_ms": [float(x) for x in po.phase_ms]}


def cycle_shares(st):
    """In-kernel s_memtime accounting of the chase kernel (psd_stats.step_cycles): share of decide / window load /
    chase / window store in the kernel's own time, and the shader clock derived from s_memrealtime."""
    c = list(st.step_cycles)
    if not c[4]:
        return None
    out = {"decide": c[0] / c[4], "window_load": c[1] / c[4], "chase": c[2] / c[4], "window_store": c[3] / c[4]}
    if c[5]:
        out["shader_clock_GHz"] = c[4] / (c[5] * 10e-9) / 1e9  # s_memrealtime ticks at 100 MHz
    return out


def pmc_traffic(n, p):
    """HBM bytes per chase launch (step + apply kernels) from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs, gfx950 correction of the guide applied; profiles/r01/pmc_traffic_cfg2.json says how).
    Counters cannot be collected from inside this process, so the figure is the one measured for this configuration
    with tools/psd_profile; None for any other configuration."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "pmc_traffic_cfg2.json")
    if (n, p) != (512, 16) or not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)["traffic_bytes_per_step_launch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--p", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    n, p = args.n, args.p
    seed = 1234 + 2  # BASELINE config index 2
    eng = psd_amd.Engine(device=local_rank)
    As = pt.bench_factors(n, p, seed)
    host = torch.from_numpy(pt.pack(As))
    total = args.steps + args.warmup
    bufs = [host.to(device) for _ in range(total)]
    zbufs = [torch.zeros_like(bufs[0]) for _ in range(total)]
    torch.cuda.synchronize()

    def run(k):
        return eng.pschur_dev(bufs[k].data_ptr(), n, p, "R", dZ_ptr=zbufs[k].data_ptr())

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

    sweeps = sum(st.nsweeps for (_, _, st, _) in results)
    elapsed_max, sweeps_all = aggregate(dist if world > 1 else None, elapsed, sweeps, device)

    if rank == 0:
        st = results[-1][2]
        lam = results[-1][0]
        # accuracy gate on the last timed step (outside the timed region)
        P = pt.product(As)
        lam_ref = np.linalg.eigvals(P)
        lam_err = pt.match_eigs(lam_ref, lam) / np.linalg.norm(P, 2)
        Ts = pt.unpack(bufs[-1].cpu().numpy())
        Zs = pt.unpack(zbufs[-1].cpu().numpy())
        ok, err = pt.checkpsd(pt.PSD(Ts, Zs, lam, "R", 1), As, thresh=100 * np.sqrt(n / 32))

        nwin = sum(s.nwindows for (_, _, s, _) in results)
        nlaunch = sum(s.nlaunch_step for (_, _, s, _) in results)
        ntrain = sum(s.reserved for (_, _, s, _) in results)
        bytes_sw = sum(s.bytes_sweeps for (_, _, s, _) in results)
        ms_iter = sum(s.ms_iter for (_, _, s, _) in results)
        ms_hess = sum(s.ms_hess for (_, _, s, _) in results)
        ms_formq = sum(s.ms_formq for (_, _, s, _) in results)
        ksamples = sum(s.step_kernel_samples for (_, _, s, _) in results)
        kms = (sum(s.step_kernel_ms_avg * s.step_kernel_samples for (_, _, s, _) in results) / ksamples) if ksamples else None
        # one launch of the chase kernel = one tick: one window of every bulge (cursor) of the running train
        bytes_per_launch = bytes_sw / max(nlaunch, 1)
        roof = None
        if kms:
            achieved = bytes_per_launch / (kms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "psd_rq_step_train", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n, p),
                    "alg_bytes_per_launch": bytes_per_launch, "avg_launch_ms": kms, "launch_samples": ksamples,
                    "windows_per_launch": nwin / max(nlaunch, 1),
                    "note": "algorithmic bytes of the sweep windows one launch chases (one window of every bulge of the "
                            "running multishift train; 2*8*p*w*(2n+1) per sweep) / HIP-event duration of the chase "
                            "kernel; every bulge is latency-bound on its serial reflector chain"}
        out = {
            "metric": "PSD sweeps/sec (pschur! n=%d p=%d Float64, Hessenberg+Q+iteration, operands in HBM)" % (n, p),
            "value": sweeps_all / elapsed_max,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: pschur!(A,:R) N=%d p=%d Float64 wantT wantZ, A_j = I + 0.5*G_j/sqrt(n)" % (n, p),
                       "seed": seed, "parallelism": "replicas x%d" % world, "window": st.window},
            "sweeps_per_step": sweeps / args.steps,
            "sweeps_in_multishift_trains_per_step": ntrain / args.steps,
            "chase_kernel_cycle_shares": cycle_shares(st),
            "phase_ms_per_step": {"hessenberg": ms_hess / args.steps, "formq": ms_formq / args.steps,
                                  "iteration": ms_iter / args.steps},
            "algorithmic_GBps": {"sweeps": bytes_sw / (ms_iter * 1e-3) / 1e9 if ms_iter else None,
                                 "hessenberg": st.bytes_hess / (st.ms_hess * 1e-3) / 1e9 if st.ms_hess else None,
                                 "whole_call": (bytes_sw / args.steps + st.bytes_hess + st.bytes_formq) * args.steps / elapsed / 1e9},
            "accuracy": {"eig_rel_err_vs_numpy_prod": lam_err, "checkpsd_ok": bool(ok), "checkpsd_max_err_eps": float(err.max())},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, p, seed)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
