"""world_size-N gloo worker for tests/test_dist_gloo.py: shards reads, "aligns" the shard with the CPU oracle (test
infrastructure standing in for the GPU path, which needs an MI355X), gathers records to rank 0 and writes them out."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from floxer_amd import distributed as D  # noqa: E402
from floxer_amd import simulate as S  # noqa: E402


def main():
    out_path = sys.argv[1]
    n_reads = int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    genome = S.make_genome(60000, 2, seed=41)
    reads, _, _ = S.make_reads(genome, n_reads, 600, 0.05, seed=42)
    reads.append(np.zeros(0, np.uint8))      # a filtered read inside the last shard
    lo, hi = D.shard_bounds(len(reads), rank, world)
    # the job's index: built by rank 0 only, every rank ends up with the same arrays
    import floxer_amd as F
    shared = D.build_index_once(genome, rank, world, device=None, tag=f"test{os.environ.get('MASTER_PORT', '')}")
    own = F.fmindex(genome)
    assert np.array_equal(shared.suffix_array_u32(), own.suffix_array_u32()) and np.array_equal(shared.bwt(True), own.bwt(True))
    assert shared.num_references == 2 and shared.device_bytes == own.device_bytes
    res = O.Index(genome).run(reads[lo:hi], O.params(error_probability=0.05))
    counts = D.exchange_counts(len(res.rows), len(res.cigars), rank, world)     # every rank learns every part's size
    assert counts.shape == (world, 2) and tuple(counts[rank]) == (len(res.rows), len(res.cigars))
    shifted = res.rows.copy()
    shifted[:, 0] += lo
    table = D.gather_rows(shifted, counts, rank, world)                          # final gather of the fixed-size records only
    merged = D.gather_records(res.rows, res.cigars, lo, rank, world)
    if rank == 0:
        assert counts[:, 0].sum() == len(merged[0]) and counts[:, 1].sum() == len(merged[1])
        got = table.numpy()
        assert got.shape[0] == counts[:, 0].sum()
        assert (got[:, :5] == merged[0][:, :5]).all() and (got[:, 6] == merged[0][:, 6]).all()
    else:
        assert table is None
    if rank == 0:
        np.savez(out_path, rows=merged[0], cigars=merged[1])
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
