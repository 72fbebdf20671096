"""GPU: bench.py as the driver launches it for N > 1 -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` --
rehearsed on ONE device: the ranks share device 0 and rendezvous over gloo (RCCL refuses two ranks on one device), the halo exchange
bounces through pinned host memory.  Everything else is the multi-rank path: per-rank shards, the exchange on its own stream
overlapped with the interior search, max-over-ranks timing, partition-independent checksums, clean exit of every rank.
(A crash of exactly this path -- the exchange stream destroyed under torch's bookkeeping -- was only visible with more than one rank.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, port):
    env = dict(os.environ, KVZ_BENCH_SHARE_DEVICE="1", KVZ_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "1",
           "--frames", "16", "--frames-4k", "16", "--search-frames", "2", "--no-cpu-baseline", "--no-encoder-leg", "--no-reference-workload"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_and_three_ranks_on_one_device_agree():
    one = _run(1, 29631)
    two = _run(2, 29632)
    three = _run(3, 29633)
    for d, n in ((one, 1), (two, 2), (three, 3)):
        assert d["n_gpus"] == n and d["value"] > 0
        sh = d["shard_4k"]
        assert "error" not in sh, sh
        assert sh["n_gpus"] == n and len(sh["share_per_rank"]) == n
        assert sh["search"]["value"] > 0
    # one fixed batch whatever the number of ranks: the checksums over all ranks do not depend on the partition
    want = one["shard_4k"]["checksums_over_all_ranks"]
    for d in (two, three):
        got = d["shard_4k"]["checksums_over_all_ranks"]
        for k in ("sum_sad", "sum_satd", "sum_abs_coeff", "sum_weighted_coeff"):
            assert got[k] == want[k], (k, got[k], want[k])
    assert two["shard_4k"]["ideal_speedup"] > 1.9 and three["shard_4k"]["ideal_speedup"] > 2.9
