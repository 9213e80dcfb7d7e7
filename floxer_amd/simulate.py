"""Synthetic genome + long-read generator with the semantics of the reference's simulator
(/root/reference/src/main/simulated_dataset.cpp:30-223), but with a portable RNG (numpy PCG64, default seed 7267281,
simulated_dataset.cpp:241) because std::uniform_int_distribution / std::sample are not reproducible across platforms.

* genome: i.i.d. uniform over {A,C,G,T} (ranks 1..4), `num_chromosomes` sequences of `chromosome_length`
* read: substring of `base_len` at a uniform start on a uniform chromosome; exactly floor(rate*base_len) distinct
  positions are mutated; kind uniform over {mismatch (always a different base), insertion (keep base, insert a uniform
  base after it), deletion}; name `id_{i}_chromosome_{c}_position_{p}_max_errors_{e}`; quality all 'I'.
* documented deviation: each read is reverse-complemented with probability `revcomp_fraction` (the reference's
  simulator only emits forward-strand reads) so that both strands are exercised.
"""
import numpy as np

DEFAULT_SEED = 7267281
_COMP = np.array([0, 4, 3, 2, 1, 5], dtype=np.uint8)


def make_genome(chromosome_length, num_chromosomes=1, seed=DEFAULT_SEED):
    rng = np.random.default_rng(seed)
    return [rng.integers(1, 5, size=chromosome_length, dtype=np.uint8) for _ in range(num_chromosomes)]


def make_reads(genome, num_reads, base_len, error_rate, seed=DEFAULT_SEED + 1, revcomp_fraction=0.5):
    """Returns (reads: list of uint8 rank arrays, names: list of str, truth: list of (chrom, pos, reverse))."""
    rng = np.random.default_rng(seed)
    num_errors = int(error_rate * base_len)
    reads, names, truth = [], [], []
    for read_id in range(num_reads):
        c = int(rng.integers(0, len(genome)))
        chrom = genome[c]
        start = int(rng.integers(0, len(chrom) - base_len))
        origin = chrom[start:start + base_len]
        mut_idx = np.sort(rng.choice(base_len, size=num_errors, replace=False))
        kinds = rng.integers(0, 3, size=num_errors)          # 0 mismatch, 1 insertion, 2 deletion
        r3 = rng.integers(0, 3, size=num_errors).astype(np.uint8)
        r4 = rng.integers(1, 5, size=num_errors).astype(np.uint8)
        keep = np.ones(base_len, dtype=bool)
        seq = origin.copy()
        mm = mut_idx[kinds == 0]
        o = (origin[mm] - 1).astype(np.uint8)
        g = r3[kinds == 0]
        seq[mm] = np.where(g >= o, g + 1, g) + 1               # choose_distinct_rank, simulated_dataset.cpp:75-79
        keep[mut_idx[kinds == 2]] = False
        ins_at = mut_idx[kinds == 1]
        ins_base = r4[kinds == 1]
        # assemble: every kept origin base, plus inserted bases after their origin base
        counts = keep.astype(np.int64)
        counts[ins_at] += 1
        out = np.empty(int(counts.sum()), dtype=np.uint8)
        ends = np.cumsum(counts)
        starts_ = ends - counts
        out[starts_[keep]] = seq[keep]
        out[ends[ins_at] - 1] = ins_base
        reverse = bool(rng.random() < revcomp_fraction)
        if reverse:
            out = _COMP[out[::-1]]
        reads.append(np.ascontiguousarray(out))
        names.append(f"id_{read_id}_chromosome_{c}_position_{start}_max_errors_{num_errors}")
        truth.append((c, start, reverse))
    return reads, names, truth


def ranks_to_str(r):
    return "".join("$ACGTN"[int(x)] for x in r)


def write_fasta(path, genome, names=None):
    with open(path, "w") as f:
        for i, g in enumerate(genome):
            f.write(f">{names[i] if names else f'chromosome_{i}'}\n")
            s = ranks_to_str(g)
            for j in range(0, len(s), 80):
                f.write(s[j:j + 80] + "\n")


def write_fastq(path, reads, names):
    with open(path, "w") as f:
        for r, n in zip(reads, names):
            f.write(f"@{n}\n{ranks_to_str(r)}\n+\n{'I' * len(r)}\n")


# ------------------------------------------------------------------------------------------------ the same, at benchmark scale
# flx_sim_genome / flx_sim_reads (floxer_amd/csrc/flx_simulate.cpp): same semantics, multi-threaded, own portable generator
# (a different stream of random numbers than make_genome / make_reads above).
def make_genome_fast(chromosome_length, num_chromosomes=1, seed=DEFAULT_SEED, repeat_rich=False):
    """one contiguous uint8 rank array of num_chromosomes * chromosome_length symbols + the list of per-chromosome views.
    repeat_rich: interspersed repeat families, tandem repeats, low complexity, runs of N and segmental duplications over about half
    of the bases (flx_sim_genome_repeats) instead of uniform random sequence"""
    import ctypes as C
    from . import capi
    pool = np.empty(chromosome_length * num_chromosomes, dtype=np.uint8)
    gen = capi.lib().flx_sim_genome_repeats if repeat_rich else capi.lib().flx_sim_genome
    capi.check(gen(len(pool), seed, capi.ptr(pool, capi.u8p)))
    return pool, [pool[i * chromosome_length:(i + 1) * chromosome_length] for i in range(num_chromosomes)]


def make_reads_fast(genome_pool, chrom_lens, num_reads, base_len, error_rate, seed=DEFAULT_SEED + 1, revcomp_fraction=0.5):
    """Returns ((pool, offsets), truth) with truth = (chrom u32[n], pos u64[n], reverse u8[n]); the (pool, offsets) pair is what
    resident_reads / aligner.align_reads take."""
    from . import capi
    lens = np.ascontiguousarray(chrom_lens, dtype=np.uint64)
    cap = num_reads * (base_len + int(error_rate * base_len))
    pool = np.empty(max(cap, 1), dtype=np.uint8)
    offs = np.zeros(num_reads + 1, dtype=np.uint64)
    chrom = np.zeros(max(num_reads, 1), dtype=np.uint32)
    pos = np.zeros(max(num_reads, 1), dtype=np.uint64)
    rev = np.zeros(max(num_reads, 1), dtype=np.uint8)
    capi.check(capi.lib().flx_sim_reads(capi.ptr(genome_pool, capi.u8p), capi.ptr(lens, capi.u64p), len(lens), num_reads, base_len,
                                        float(error_rate), float(revcomp_fraction), seed, capi.ptr(pool, capi.u8p), len(pool),
                                        capi.ptr(offs, capi.u64p), capi.ptr(chrom, capi.u32p), capi.ptr(pos, capi.u64p),
                                        capi.ptr(rev, capi.u8p)))
    return (pool[: int(offs[-1])], offs), (chrom[:num_reads], pos[:num_reads], rev[:num_reads])
