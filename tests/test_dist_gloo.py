"""CPU tier: the N>1 bench path (one process per GPU, barrier + max-over-ranks timing, replica aggregation)
rehearsed with gloo, world_size 2."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_aggregation_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch.distributed as dist\n"
        "import bench\n"
        "dist.init_process_group('gloo')\n"
        "r = dist.get_rank()\n"
        "t, units = bench.aggregate(dist, seconds=1.0 + r, units=10 * (r + 1), device=None)\n"
        "sl = bench.period_slice(16, dist.get_world_size(), r)\n"
        "if r == 0: print(json.dumps({'t': t, 'units': units, 'slice': sl}))\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json

    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["t"] == 2.0 and res["units"] == 30 and res["slice"] == [0, 8]


_SHARD_WORKER = r'''
import sys, os, json
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch.distributed as dist
import psd_amd, psdtest as pt
import importlib.util  # (the package directory name contains a dot: load the module by path, as psd_amd.py does)
spec = importlib.util.spec_from_file_location("psd_sharded", os.path.join(ROOT, "periodicschurdecompositions.jl_amd", "sharded.py"))
sharded = importlib.util.module_from_spec(spec); spec.loader.exec_module(sharded)

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# TEST-ONLY serial simulation of the device code (tests/hostsim): one engine per rank, as one process per GPU would have
eng = psd_amd.Engine(libpath=os.path.join(ROOT, "tests", "hostsim", "_build", "libpsd_hostsim.so"))
out = {}
for case in CASES:
    n, p, lr = case[:3]
    cplx = len(case) > 3 and case[3] == "c"  # ComplexF64 (psd_z_pschur: BASELINE configs[2])
    A = pt.bench_factors(n, p, seed=77 + n + p, dtype=(np.complex128 if cplx else np.float64))
    ref = eng.pschur(A, lr)                               # one rank holding everything
    part = sharded.pschur_sharded(eng, A, lr, dist=dist, gather=False)
    owned = part.owned
    assert owned == list(eng_owned(eng, p, lr, rank, world)), (owned,)
    # the chains and the factors are replicated: bit-identical to the unsharded run on every rank
    assert all(np.array_equal(a, b) for a, b in zip(ref.Ts, part.Ts))
    assert np.array_equal(ref.values, part.values)
    # owned Schur vectors are the unsharded ones, bit for bit; the others were not computed here
    for j in range(p):
        if owned[j]:
            assert np.array_equal(ref.Z[j], part.Z[j]), j
    full = sharded.pschur_sharded(eng, A, lr, dist=dist, gather=True)
    assert all(np.array_equal(a, b) for a, b in zip(ref.Z, full.Z))
    ok, err = pt.checkpsd(full, A, thresh=100 * np.sqrt(max(n / 32, 1)))
    assert ok, err
    P = pt.product(A, left=(lr == "L"))
    assert pt.match_eigs(np.linalg.eigvals(P), full.values) <= 1e-10 * np.linalg.norm(P, 2)
    if cplx:
        assert all(np.iscomplexobj(z) and np.abs(z.imag).max() > 0 for z in full.Z)  # (the gather keeps the imaginary parts)
    out[f"{n}x{p}{lr}" + ("c" if cplx else "")] = {"owned": int(sum(owned)), "resid": float(err.max())}
if rank == 0:
    print(json.dumps(out))
dist.destroy_process_group()
'''


def _run_sharded(tmp_path, world, cases, port):
    script = tmp_path / "shard_worker.py"
    head = (f"ROOT = {ROOT!r}\nCASES = {cases!r}\n"
            "def eng_owned(eng, p, lr, rank, world):\n"
            "    eng.set_shard(rank, world)\n"
            "    o = eng.owned_slots(p, lr)\n"
            "    eng.set_shard(0, 1)\n"
            "    return [bool(x) for x in o]\n")
    script.write_text(head + _SHARD_WORKER)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    import json

    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_period_sharded_pschur_gloo_world2(tmp_path):
    """The period-sharded engine (psd_set_shard: Z_j of a contiguous slice of the period per rank, chains replicated,
    one all-gather of the slices at the end) on two ranks: same decomposition as one rank, bit for bit, both
    orientations, a period the ranks split evenly and one they do not."""
    res = _run_sharded(tmp_path, 2, [(24, 4, "R"), (30, 5, "L"), (40, 3, "R"), (70, 4, "L"), (26, 5, "R", "c"), (22, 4, "L", "c")], 29541)  # (n >= 64: blocked Q formation of a slice)
    assert res["24x4R"]["owned"] == 2 and res["30x5L"]["owned"] == 3 and res["40x3R"]["owned"] == 2
    assert res["26x5Rc"]["owned"] == 3 and res["22x4Lc"]["owned"] == 2


def test_period_sharded_pschur_gloo_world3(tmp_path):
    res = _run_sharded(tmp_path, 3, [(20, 7, "R"), (18, 2, "L"), (66, 5, "R")], 29543)
    assert res["20x7R"]["owned"] == 3 and res["18x2L"]["owned"] == 1


_RING_WORKER = r'''
import sys, os, json
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, torch.distributed as dist
import psdtest as pt, ring_rehearsal as rr

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, p = NP
A = pt.bench_factors(n, p, seed=31 + n + p, dtype=np.complex128)
H0 = [np.triu(a, -1 if j == 0 else 0) for j, a in enumerate(A)]   # Hessenberg-triangular input (test/generalized.jl:67-76)
lo, hi = rr.period_slice(p, world, rank)
# this rank keeps ONLY its slice of the factors and of the Schur vectors
Hloc = [H0[j].copy() for j in range(lo, hi)]
Zloc = [np.eye(n, dtype=np.complex128) for _ in range(lo, hi)]
resident = sum(h.nbytes for h in Hloc) + sum(z.nbytes for z in Zloc)
ring = rr.Ring(dist)
lam, st = rr.ring_pschur_hess(Hloc, Zloc, n, p, ring)
# --- for the check only: everything to rank 0
def gather(loc):
    per = max(rr.period_slice(p, world, r)[1] - rr.period_slice(p, world, r)[0] for r in range(world))
    send = torch.zeros((per, n, n), dtype=torch.complex128)
    for k, m in enumerate(loc):
        send[k] = torch.from_numpy(np.ascontiguousarray(m))
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    out = []
    for r in range(world):
        l, h = rr.period_slice(p, world, r)
        out += [recv[r][k].numpy() for k in range(h - l)]
    return out
T, Z = gather(Hloc), gather(Zloc)
busy = torch.tensor([st["busy_beats"] / max(st["beats"], 1)], dtype=torch.float64)
bl = [torch.zeros_like(busy) for _ in range(world)]
dist.all_gather(bl, busy)
if rank == 0:
    P = pt.product(H0)
    nP = np.linalg.norm(P, 2)
    err = pt.match_eigs(np.linalg.eigvals(P), lam) / nP
    res = max(np.linalg.norm(H0[l] - Z[l] @ T[l] @ Z[(l + 1) % p].conj().T) / (pt.EPS * np.linalg.norm(H0[l], 1)) for l in range(p))
    orth = max(np.linalg.norm(Z[l] @ Z[l].conj().T - np.eye(n)) / (pt.EPS * n) for l in range(p))
    tri = max(np.abs(np.tril(T[l], -1)).max() for l in range(p))
    # the one-rank run of the same code on the same input: same spectrum
    H1 = [h.copy() for h in H0]; Z1 = [np.eye(n, dtype=np.complex128) for _ in range(p)]
    lam1, st1 = rr.ring_pschur_hess(H1, Z1, n, p, rr.Ring(None))
    same = pt.match_eigs(lam1, lam) / nP
    print(json.dumps({"world": world, "eig_err": err, "resid_eps": res, "orth_eps_n": orth, "tri": tri, "vs_one_rank": same,
                      "resident_bytes_per_rank": resident, "whole_problem_bytes": 2 * p * n * n * 16,
                      "p2p_messages": ring.p2p_messages, "p2p_bytes": ring.p2p_bytes, "collectives": ring.collectives,
                      "sweeps": st["sweeps"], "trains": st["trains"], "busy_fraction_by_rank": [float(b.item()) for b in bl]}))
dist.destroy_process_group()
'''


def _run_ring(tmp_path, world, n, p, port):
    script = tmp_path / f"ring{world}.py"
    script.write_text(f"ROOT = {ROOT!r}\nNP = ({n}, {p})\n" + _RING_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    import json

    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_north_star_partition_ring_world2(tmp_path):
    """The north_star partition: factors AND Schur vectors sharded by period, rotation hand-off at the slice
    boundaries point to point, a train of bulges in flight, the scalar product data by slice partials + all-gather:
    tests/ring_rehearsal.py on 2 gloo ranks reproduces the spectrum and the reference's checkpsd invariants
    (diagnostics.jl:190-263) with half of the problem resident per rank."""
    r = _run_ring(tmp_path, 2, 36, 6, 29541)
    assert r["eig_err"] <= 1e-10 and r["vs_one_rank"] <= 1e-10
    assert r["resid_eps"] <= 100 and r["orth_eps_n"] <= 10 and r["tri"] == 0.0
    assert r["resident_bytes_per_rank"] * 2 == r["whole_problem_bytes"]
    assert r["p2p_messages"] > 0 and min(r["busy_fraction_by_rank"]) > 0.3


def test_north_star_partition_ring_world3_ragged(tmp_path):
    r = _run_ring(tmp_path, 3, 30, 7, 29543)  # (slices of 3, 2, 2 factors)
    assert r["eig_err"] <= 1e-10 and r["vs_one_rank"] <= 1e-10
    assert r["resid_eps"] <= 100 and r["orth_eps_n"] <= 10 and r["tri"] == 0.0
    assert r["resident_bytes_per_rank"] * 7 == r["whole_problem_bytes"] * 3
