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
