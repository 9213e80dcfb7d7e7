"""The N>1 path (read sharding + gather of alignment records to rank 0 + global read order) on CPU with gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from floxer_amd import distributed as D
from floxer_amd import simulate as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_reads_in_order():
    for n in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) <= -(-n // world) if n else True


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_equals_single_process(tmp_path, world):
    n_reads = 11
    out = str(tmp_path / f"merged_{world}.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(n_reads)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got = np.load(out)
    genome = S.make_genome(60000, 2, seed=41)
    reads, _, _ = S.make_reads(genome, n_reads, 600, 0.05, seed=42)
    reads.append(np.zeros(0, np.uint8))
    exp = O.Index(genome).run(reads, O.params(error_probability=0.05))
    got_recs = O.RunResult(got["rows"], got["cigars"], None, None, 0).records()
    assert got_recs == exp.records()
