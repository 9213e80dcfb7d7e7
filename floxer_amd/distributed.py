"""Multi-GPU read sharding: reads are independent units (parallelization.cpp:77-87), so each rank
aligns a contiguous block of the reads against its own replica of the index; there is no collective on the data path.

Output is in input order, so the job's output is the concatenation of the ranks' parts in rank order. Two ways to get there:
  * `exchange_counts` (what `bench.py` times): every rank keeps (writes) its own part and the ranks exchange only the sizes of
    their parts, i.e. the offsets at which the parts are concatenated. Without `-I` a 5-kb read yields ~32 KB of CIGAR words, so
    funnelling all parts through one rank would make that rank's host the bottleneck of the whole job.
  * `gather_records`: one variable-length gather of all records to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo on
    CPU in tests) for callers that want a single stream; rank 0 re-emits the records in global read order, so the result is
    identical for any number of ranks."""
import numpy as np


def shard_bounds(n_reads, rank, world):
    """contiguous blocks of ceil(N / G) reads in input order"""
    per = -(-n_reads // world) if world > 0 else n_reads
    lo = min(n_reads, rank * per)
    return lo, min(n_reads, lo + per)


def exchange_counts(n_records, n_cigar_words, rank, world, device=None):
    """all-gather of every part's (records, CIGAR words). Returns an int64 array (world, 2): row r = sizes of rank r's part; the
    running sums are the offsets of the parts in the job's output."""
    if world == 1:
        return np.array([[n_records, n_cigar_words]], dtype=np.int64)
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    mine = torch.tensor([n_records, n_cigar_words], device=dev, dtype=torch.int64)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return torch.stack(parts).cpu().numpy()


def gather_rows(rows, counts, rank, world, device=None):
    """Final gather of the fixed-size alignment records (read, flag, reference, position, NM, CIGAR offset/length in the owner's
    part) to rank 0 — the job's mapping table; the CIGAR words stay in the owners' parts. rows: (n,7) int64 of this rank (read
    index already global); counts: exchange_counts-style (world, >=1) array whose column 0 is every rank's number of rows.
    Returns on rank 0 a list of `world` tensors (left on `device`: over RCCL nothing passes through a host), None elsewhere."""
    import torch
    rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1, 7)
    if world == 1:
        return [torch.from_numpy(rows)]
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    n_max = max(1, int(np.max(np.asarray(counts)[:, 0])))
    mine = torch.zeros((n_max, 7), device=dev, dtype=torch.int64)
    mine[: rows.shape[0]] = torch.from_numpy(rows).to(dev)
    if rank == 0:
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.gather(mine, parts, dst=0)
        return [parts[r][: int(counts[r][0])] for r in range(world)]
    dist.gather(mine, None, dst=0)
    return None


def gather_records(rows, cigars, read_offset, rank, world, device=None):
    """rows: (n,7) int64 {read_index (shard-local), flag, ref_id, pos, nm, cigar_off, cigar_len}; cigars: uint32 words.
    Returns (rows, cigars) of the whole job on rank 0 (read_index global, cigar offsets rebased), None on other ranks."""
    rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1, 7).copy()
    rows[:, 0] += read_offset
    cigars = np.ascontiguousarray(cigars, dtype=np.uint32)
    if world == 1:
        return rows, cigars
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    t_rows = torch.from_numpy(rows).to(dev)
    t_cig = torch.from_numpy(cigars.view(np.int32)).to(dev)          # the same 32 bits; torch has no uint32 collectives
    counts = torch.tensor([t_rows.shape[0], t_cig.shape[0]], device=dev, dtype=torch.int64)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    all_counts = [(int(c[0].item()), int(c[1].item())) for c in all_counts]
    max_r = max(1, max(c[0] for c in all_counts))
    max_c = max(1, max(c[1] for c in all_counts))
    pad_r = torch.zeros((max_r, 7), device=dev, dtype=torch.int64)
    pad_r[: t_rows.shape[0]] = t_rows
    pad_c = torch.zeros((max_c,), device=dev, dtype=torch.int32)
    pad_c[: t_cig.shape[0]] = t_cig
    if rank == 0:
        gr = [torch.zeros_like(pad_r) for _ in range(world)]
        gc = [torch.zeros_like(pad_c) for _ in range(world)]
        dist.gather(pad_r, gr, dst=0)
        dist.gather(pad_c, gc, dst=0)
        out_rows, out_cig, base = [], [], 0
        for r in range(world):
            nr, nc = all_counts[r]
            rr = gr[r][:nr].cpu().numpy().copy()
            rr[:, 5] += base
            out_rows.append(rr)
            out_cig.append(gc[r][:nc].cpu().numpy().view(np.uint32))
            base += nc
        return np.concatenate(out_rows, axis=0), np.concatenate(out_cig)
    dist.gather(pad_r, None, dst=0)
    dist.gather(pad_c, None, dst=0)
    return None
