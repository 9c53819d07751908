"""Diagnostic (not the bench line): B independent pschur!(A,:R) problems of the bench configuration solved concurrently
on ONE GPU, one engine context (own stream, own workspace) per host thread.  A single decomposition keeps one compute
unit busy with its chase, so independent problems scale almost linearly until the bulk-apply kernels and the host
threads saturate.  Prints one JSON line per B."""
import json
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402

torch.cuda.init()
import psd_amd  # noqa: E402
import psdtest as pt  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Bs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16]
As = pt.bench_factors(n, p, 1236)
host = torch.from_numpy(pt.pack(As))
dev = torch.device("cuda:0")
for B in Bs:
    engs = [psd_amd.Engine() for _ in range(B)]
    bufs = [[host.to(dev) for _ in range(2)] for _ in range(B)]
    zb = [[torch.zeros_like(bufs[0][0]) for _ in range(2)] for _ in range(B)]
    torch.cuda.synchronize()
    res = [None] * B

    def work(b, k):
        res[b] = engs[b].pschur_dev(bufs[b][k].data_ptr(), n, p, "R", dZ_ptr=zb[b][k].data_ptr())

    for k in range(2):  # k = 0 warm-up, k = 1 timed
        th = [threading.Thread(target=work, args=(b, k)) for b in range(B)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    sweeps = sum(r[2].nsweeps for r in res)
    lam = res[-1][0]
    lam_ref = np.linalg.eigvals(pt.product(As))
    print(json.dumps({"concurrent_problems": B, "n": n, "p": p, "seconds": el, "sweeps_per_s": sweeps / el,
                      "problems_per_s": B / el, "eig_rel_err": pt.match_eigs(lam_ref, lam) / np.linalg.norm(pt.product(As), 2)}),
          flush=True)
    del engs, bufs, zb
