"""GPU scan of the 2 x 2 standardisation weak spot: the ensemble of tools/micro/repro_scan.py, with
the CPU oracle's verdict beside every device failure.  Usage: python tools/r04/scan2x2.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
torch.cuda.init()
import psd_amd, psdtest as pt
eng = psd_amd.Engine()
bad = []; tot = 0
for n in [4, 5, 6, 7, 8, 9, 10, 12, 16, 20]:
    for p in [8, 9, 16, 39, 64]:
        for lr in "RL":
            for eps_ in [0.3, 0.8]:
                for seed in [1, 2, 3, 658499203]:
                    A = pt.bench_factors(n, p, seed=seed, eps=eps_)
                    ps = eng.pschur(A, lr)
                    th = 100 * np.sqrt(max(n / 32, 1))
                    ok, err = eng.checkpsd(ps, A, thresh=th)
                    tot += 1
                    if not ok:
                        po = pt.oracle_pschur(A, lr)
                        oko, erro = pt.checkpsd(po, A, thresh=th)
                        rec = dict(n=n, p=p, lr=lr, eps=eps_, seed=seed, dev_err=float(np.max(err)), oracle_ok=bool(oko), oracle_err=float(np.max(erro)))
                        bad.append(rec)
                        print("FAIL", rec, flush=True)
out = dict(tag=os.environ.get("TAG", ""), c2=os.environ.get("PSD_C2", "default"), total=tot, failures=bad,
           device_only=[r for r in bad if r["oracle_ok"]])
print(json.dumps(dict(total=tot, bad=len(bad), device_only=len(out["device_only"]))))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
