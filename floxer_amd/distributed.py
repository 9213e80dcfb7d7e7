"""Multi-GPU read sharding: reads are independent units (parallelization.cpp:77-87), so each rank
aligns a contiguous block of the reads against its own replica of the index; there is no collective on the data path.

Output is in input order, so the job's output is the concatenation of the ranks' parts in rank order. Two ways to get there:
  * `exchange_counts` (what `bench.py` times): every rank keeps (writes) its own part and the ranks exchange only the sizes of
    their parts, i.e. the offsets at which the parts are concatenated. Without `-I` a 5-kb read yields ~32 KB of CIGAR words, so
    funnelling all parts through one rank would make that rank's host the bottleneck of the whole job.
  * `gather_records`: one variable-length gather of all records to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo on
    CPU in tests) for callers that want a single stream; rank 0 re-emits the records in global read order, so the result is
    identical for any number of ranks.
Both gathers are true gathervs (all-gather of the counts, then grouped send/recv of exactly the counted rows)."""
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


def _gatherv(mine, counts, rank, world):
    """variable-length gather to rank 0 of one tensor per rank whose first dimension is counts[r] on rank r: the receiver posts one
    receive of exactly counts[r] rows per sender straight into its slice of the result, the senders one send each, all in one
    group (RCCL: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI; gloo: its send/recv pairs). Nothing is padded to
    the largest part and nothing is copied after it has arrived. Returns the concatenation on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    counts = [int(c) for c in counts]
    if rank == 0:
        out = torch.empty((sum(counts),) + tuple(mine.shape[1:]), device=mine.device, dtype=mine.dtype)
        out[: counts[0]] = mine
        ops, at = [], counts[0]
        for r in range(1, world):
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, out[at: at + counts[r]], r))
            at += counts[r]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out
    if counts[rank]:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, mine, 0)]):
            w.wait()
    return None


def gather_rows(rows, counts, rank, world, device=None):
    """Final gather of the fixed-size alignment records (read, flag, reference, position, NM, CIGAR offset/length in the owner's
    part) to rank 0 — the job's mapping table; the CIGAR words stay in the owners' parts. rows: (n,7) int64 of this rank (read
    index already global); counts: exchange_counts-style (world, >=1) array whose column 0 is every rank's number of rows.
    Returns on rank 0 one (sum of counts, 7) tensor in rank order (left on `device`: over RCCL nothing passes through a host),
    None elsewhere."""
    import torch
    rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1, 7)
    if world == 1:
        return torch.from_numpy(rows)
    dev = device if device is not None else torch.device("cpu")
    return _gatherv(torch.from_numpy(rows).to(dev), np.asarray(counts)[:, 0], rank, world)


def gather_records(rows, cigars, read_offset, rank, world, device=None):
    """rows: (n,7) int64 {read_index (shard-local), flag, ref_id, pos, nm, cigar_off, cigar_len}; cigars: uint32 words.
    Returns (rows, cigars) of the whole job on rank 0 (read_index global, cigar offsets rebased), None on other ranks."""
    rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1, 7).copy()
    rows[:, 0] += read_offset
    cigars = np.ascontiguousarray(cigars, dtype=np.uint32)
    if world == 1:
        return rows, cigars
    import torch
    dev = device if device is not None else torch.device("cpu")
    counts = exchange_counts(len(rows), len(cigars), rank, world, device=device)
    # a part's CIGAR offsets are rebased by the words of the parts in front of it before it leaves its owner
    rows[:, 5] += int(counts[:rank, 1].sum())
    t_rows = _gatherv(torch.from_numpy(rows).to(dev), counts[:, 0], rank, world)
    t_cig = _gatherv(torch.from_numpy(cigars.view(np.int32)).to(dev), counts[:, 1], rank, world)   # the same 32 bits; no uint32 collectives
    if rank == 0:
        return t_rows.cpu().numpy(), t_cig.cpu().numpy().view(np.uint32)
    return None


def replicate_index(references, rank, world, device):
    """The FM index of a multi-rank job, replicated per GPU: rank 0 builds it (suffix arrays, BWTs, occurrence tables
    on its GPU) and uploads its HBM image into five device buffers; the buffers go to every other rank's HBM with RCCL broadcasts
    over xGMI, the small host part (sequence starts / lengths, symbol counts) with an object broadcast. No rank but the first ever
    holds the index in host memory, builds it or reads a file. Returns (index, image): `context(index, device, image=image)`.
    world == 1: (index, None)."""
    import floxer_amd as F
    if world == 1:
        return F.fmindex(references, device=device), None
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", device)
    index, payload = None, [None]
    if rank == 0:
        index = F.fmindex(references, device=device)
        payload = [(index.meta(), index.image_layout())]
    dist.broadcast_object_list(payload, src=0)
    meta, sizes = payload[0]
    image = [torch.empty(int(n), dtype=torch.uint8, device=dev) for n in sizes]
    if rank == 0:
        index.image_upload(device, [b.data_ptr() for b in image])
    for b in image:
        dist.broadcast(b, src=0)
    if rank != 0:
        index = F.fmindex.from_meta(meta)
    return index, image


def build_index_once(references, rank, world, device=None, tag=None):
    """The FM index of a multi-rank job: built once (rank 0; suffix arrays / BWTs / occurrence tables on HIP device `device`, on the
    host when it is None), written to a file in shared memory, loaded by the other ranks, removed again. world == 1: just built.
    Every rank then uploads its replica to its own GPU (flx_ctx_create)."""
    import os
    import floxer_amd as F
    if world == 1:
        return F.fmindex(references, device=device)
    import torch.distributed as dist
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path = os.path.join(base, f"flx_index_{tag or os.environ.get('MASTER_PORT', 'job')}_{os.getuid()}.bin")
    index = None
    try:
        if rank == 0:
            index = F.fmindex(references, device=device)
            index.save(path)
        dist.barrier()
        if rank != 0:
            index = F.fmindex(path=path)
        dist.barrier()
    finally:
        if rank == 0 and os.path.exists(path):
            os.remove(path)
    return index
