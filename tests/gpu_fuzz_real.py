"""Randomised sweep of the real standard path (multi-block scheduler, trains with tight cursor spacing, work-list apply,
look-ahead Hessenberg) on the GPU, or with --sim on the simulated tier: random orders up to --nmax, periods up to 40,
both orientations; judged with checkpsd (on the device) and the eigenvalues of the explicit product.
Prints one line per failure and a JSON summary."""
import json
import os
import sys
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
SIM = "--sim" in sys.argv
if not SIM:
    import torch

    torch.cuda.init()
import psd_amd  # noqa: E402
import psdtest as pt  # noqa: E402


def arg(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


budget_s, nmax = arg("--seconds", 200.0), arg("--nmax", 400)
eng = psd_amd.Engine(libpath=os.path.join(HERE, "hostsim", "_build", "libpsd_hostsim.so")) if SIM else psd_amd.Engine()
rng = np.random.default_rng(arg("--seed", 20261004))
t0, ncase, fails, worst, reflimit = time.time(), 0, [], 0.0, 0
while time.time() - t0 < budget_s:
    ncase += 1
    n = int(rng.integers(2, nmax + 1))
    p = int(rng.integers(1, 41))
    lr = "RL"[int(rng.integers(0, 2))]
    seed = int(rng.integers(1, 1 << 30))
    eps_ = float(rng.choice([0.3, 0.5, 0.8]))
    tag = (n, p, lr, eps_, seed)
    try:
        A = pt.bench_factors(n, p, seed=seed, eps=eps_)
        ps = eng.pschur(A, lr)
        thresh = 100 * max(1.0, np.sqrt(n / 32)) * (1.0 if eps_ <= 0.5 else 2.0)
        if SIM:
            ok, err = pt.checkpsd(ps, A, thresh=thresh)
        else:
            ok, err = eng.checkpsd(ps, A, thresh=thresh)
        if not ok and n <= 200:
            # the reference algorithm itself misses its checkpsd on some inputs (2x2 standardisation of widely split
            # real pairs): the CPU restatement decides whether this is such a case
            po = pt.oracle_pschur(A, lr)
            oko, erro = pt.checkpsd(po, A, thresh=thresh)
            if not oko and float(err.max()) <= 4.0 * float(erro.max()):
                reflimit += 1
                continue
        worst = max(worst, float(err.max()) / thresh)
        assert ok, ("checkpsd", float(err.max()), thresh)
        P = pt.product(A, left=(lr == "L"))
        sc = np.linalg.norm(P, 2)
        lam = np.linalg.eigvals(P)
        d = pt.match_eigs(lam, ps.values) if n <= 160 else abs(np.sort(np.abs(lam)) - np.sort(np.abs(ps.values))).max()
        assert d <= 1e-9 * sc * max(1.0, np.linalg.cond(P) * 1e-4), ("eigs", d, sc)
        assert abs(np.sum(lam) - np.sum(ps.values)) <= 1e-9 * sc * n
    except Exception as e:  # noqa: BLE001
        fails.append({"case": tag, "error": repr(e)[:300]})
        print("FAIL", tag, repr(e)[:300], flush=True)
        if len(fails) > 20:
            break
print(json.dumps({"cases": ncase, "failures": len(fails), "seconds": time.time() - t0, "nmax": nmax,
                  "worst_resid_over_thresh": worst, "reference_limit_cases": reflimit, "fails": fails[:5]}))
sys.exit(1 if fails else 0)
