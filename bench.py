#!/usr/bin/env python3
"""bench.py — reads/s of the seed-and-verify hot path on MI355X. Default: the configuration BASELINE.json's metric is quoted on,
10 kb reads @ 8 % error against a GRCh38-size reference (3.1 Gb in 25 sequences; synthetic, uniform over ACGT: hg38 itself is not
available offline), floxer's default flags. `--config` selects the other configurations of BASELINE.json (E. coli size / 5 kb,
chr1 size / 10 kb, GRCh38 size / 20 kb @ 2 %), `--repeat-rich` a reference with a human-like repeat content instead of the uniform one.

One "step" = one pass of the whole path (PEX seeding -> FM search -> hierarchical verification -> root alignment with CIGAR ->
records) over one batch of synthetic long reads that is already resident in HBM. Reads shard across ranks with no data-path
collective (FM index replicated per GPU). Every rank keeps its part of the output; per step the ranks exchange the sizes of their
parts (RCCL all-gather) and the step's fixed-size alignment records go to rank 0 with a gatherv (RCCL over xGMI) - inside the timed
region, while the following steps compute (the gather streams per batch). Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W
  python bench.py --gpus N ...                 (spawns its N ranks itself, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The lanes of a context are HIP streams; ROCm multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and kernels of streams that share a queue do not overlap. Read by the HIP runtime when it initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_TAG = "r04"        # profiles/<tag>_pmc_traffic_<kernel>.json, profiles/<tag>_oracle_extensions.json

# BASELINE.json configs (per GPU): reference length, sequences, read length, error rate, reads per step
CONFIGS = {
    "grch38": dict(genome=3_100_000_000, chromosomes=25, read_length=10000, error_rate=0.08, reads=16384,
                   name="BASELINE.json configs[3] shape per GPU, the metric's configuration"),
    "chr1": dict(genome=248_000_000, chromosomes=1, read_length=10000, error_rate=0.08, reads=16384, name="BASELINE.json configs[2] shape"),
    "ecoli": dict(genome=4_600_000, chromosomes=1, read_length=5000, error_rate=0.08, reads=32768, name="BASELINE.json configs[1] shape"),
    "hifi": dict(genome=3_100_000_000, chromosomes=25, read_length=20000, error_rate=0.02, reads=8192, name="BASELINE.json configs[4] shape per GPU"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="grch38", help="workload shape (BASELINE.json configs); the flags below override its values")
    ap.add_argument("--repeat-rich", action="store_true", help="reference with interspersed repeat families, tandem repeats, low complexity, N runs and "
                    "segmental duplications over half of its bases (flx_sim_genome_repeats) instead of uniform random sequence")
    ap.add_argument("--reads-per-step", type=int, default=int(os.environ.get("FLX_BENCH_READS", 0)), help="per GPU; 0 = the config's")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("FLX_LANES", 0)),
                    help="concurrent lanes (stream + host thread) per GPU; 0 = 16 (8 on fewer than 4 cores per rank)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("FLX_BENCH_INFLIGHT", 3)),
                    help="steps submitted to the context at a time (host threads calling align_reads); every step still runs in "
                         "full inside the timed region")
    ap.add_argument("--no-isolated-pass", action="store_true", help="skip the one-lane instrumented pass (timeline profiling)")
    ap.add_argument("--isolated-only", action="store_true",
                    help="only the one-lane instrumented pass (used under rocprofv3 so that its per-kernel averages are those of roofline)")
    ap.add_argument("--genome", type=int, default=0, help="synthetic reference length in total; 0 = the config's")
    ap.add_argument("--chromosomes", type=int, default=0, help="sequences the reference is cut into; 0 = the config's")
    ap.add_argument("--read-length", type=int, default=0)
    ap.add_argument("--error-rate", type=float, default=-1.0)
    ap.add_argument("--interval-optimization", action="store_true")
    ap.add_argument("--gather-cigars", action="store_true", help="N > 1: the per-step gather to rank 0 also moves the CIGAR words (default: the fixed-size records)")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("FLX_BENCH_CPU_SAMPLE", 768)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-gathered", default="", help="rank 0 saves the record table of timed step 0 as it arrives from all ranks (numpy .npy; tests)")
    ap.add_argument("--no-host-inputs-leg", action="store_true", help="skip the second timed region (reads in pageable host memory: H2D and Peq build inside the clock)")
    ap.add_argument("--no-repeat-rich-leg", action="store_true", help="metric configuration only: skip the secondary line on the repeat-rich (hg38-like) reference")
    ap.add_argument("--repeat-rich-steps", type=int, default=8)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.genome = args.genome or cfg["genome"]
    args.chromosomes = args.chromosomes or cfg["chromosomes"]
    args.read_length = args.read_length or cfg["read_length"]
    args.error_rate = cfg["error_rate"] if args.error_rate < 0 else args.error_rate
    args.reads_per_step = args.reads_per_step or cfg["reads"]
    return args


def usable_cores():
    """threads the host actually grants this process: cgroup cpu.max quota if set, else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def compare_with_oracle(got, exp, n_sample):
    """records of the reads [0, n_sample) of a product result against an oracle run over exactly those reads: same number of records,
    and record by record (in output order) the same read, flag, reference, position, edit distance and CIGAR words"""
    import numpy as np
    g = got.rows[got.rows[:, 0] < n_sample]
    e = exp.rows
    out = {"reads": int(n_sample), "records": int(len(e)), "product_records": int(len(g)), "equal": False}
    if len(g) != len(e):
        return out
    if len(e) and not np.array_equal(g[:, :5], e[:, :5]):
        out["first_difference"] = int(np.nonzero((g[:, :5] != e[:, :5]).any(axis=1))[0][0])
        return out
    if not np.array_equal(g[:, 6], e[:, 6]):
        return out
    # CIGAR words: gather both sides' slabs in record order and compare once
    def words(rows, cig):
        if not len(rows):
            return np.zeros(0, np.uint32)
        lens = rows[:, 6]
        idx = np.repeat(rows[:, 5] - np.concatenate(([0], np.cumsum(lens)[:-1])), lens) + np.arange(int(lens.sum()))
        return np.asarray(cig, dtype=np.uint32)[idx]
    out["cigar_words"] = int(e[:, 6].sum())
    out["equal"] = bool(np.array_equal(words(g, got.cigars), words(e, exp.cigars)))
    return out


def log(msg):
    if os.environ.get("FLX_BENCH_VERBOSE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(n):
    """`bench.py --gpus N` started without a launcher: this process (which has not touched the GPU: nothing below the argument parser has
    run) starts the N ranks as children, one per GPU, with the rendezvous on 127.0.0.1, and leaves with the worst of their exit codes.
    Rank 0's JSON line is the children's only stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    cores = max(1, usable_cores() // max(1, local_world))
    os.environ.setdefault("FLX_SIM_THREADS", str(cores))
    if args.lanes <= 0:
        # A lane is a stream and a host thread that sleeps while its chunk is on the GPU: the number of lanes is what the GPU needs to
        # have chunks in every stage (16), not the number of cores (measured with 16 lanes: 16 cores 81 k reads/s, 4 cores 77.6 k,
        # 3 cores 73.8 k, 2 cores 66.5 k; 8 lanes on 16 cores: 72.6 k); below 4 cores per rank the lanes' host threads start to queue
        args.lanes = 16 if cores >= 4 else 8

    import numpy as np
    import torch
    import torch.distributed as dist
    import floxer_amd as F
    from floxer_amd import simulate as S
    from floxer_amd import distributed as D

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # FLX_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (the ranks then share
    # devices; gloo stages device tensors through the host). The measured configuration is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("FLX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- workload
    t0 = time.time()
    chrom_len = args.genome // args.chromosomes
    pool, genome = S.make_genome_fast(chrom_len, args.chromosomes, seed=S.DEFAULT_SEED, repeat_rich=args.repeat_rich)
    chrom_lens = [chrom_len] * args.chromosomes
    log(f"genome {time.time() - t0:.1f} s")
    # Every step has its own batch of reads, resident in HBM before the clock starts (0.57 GB per batch with its Peq planes) - up
    # to FLX_BENCH_MAX_BATCHES (48) of them: a longer run goes through the timed batches again in turn (nothing of an earlier pass
    # over a batch is kept but its Peq planes and its 2-bit form, 0.3 ms of kernel time per step), so that --steps 200 does not ask
    # for 120 GB of reads next to the index and the lanes' workspaces.
    n_timed_batches = min(args.steps, int(os.environ.get("FLX_BENCH_MAX_BATCHES", 48)))
    n_batches = n_timed_batches + args.warmup
    B = args.reads_per_step
    t0 = time.time()
    # every rank and every step gets its own reads (weak scaling: per-GPU work is fixed)
    batches = [S.make_reads_fast(pool, chrom_lens, B, args.read_length, args.error_rate, seed=S.DEFAULT_SEED + 1 + rank * 1000 + b)[0]
               for b in range(max(1, n_batches))]
    log(f"reads {time.time() - t0:.1f} s")

    os.environ["FLX_LANES"] = str(args.lanes)
    t0 = time.time()
    # the index is built once per job, on rank 0's GPU; its HBM image reaches the other ranks' HBM by RCCL broadcast over xGMI
    # (index replicated per GPU). Not timed (floxer's stopwatch excludes it too, floxer.cpp:154)
    index, image = D.replicate_index(genome, rank, world, local_rank)
    index_s = time.time() - t0
    log(f"index {index_s:.1f} s")
    ctx = F.context(index, device=local_rank, image=image)
    p = F.params(error_probability=args.error_rate, interval_optimization=args.interval_optimization)
    al = F.aligner(ctx, p)
    resident = [F.resident_reads(ctx, r) for r in batches]       # inputs resident in HBM before the timed region
    log("reads resident")
    first_timed = args.warmup if len(batches) > args.warmup else 0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_records = 0
    elapsed = 1.0
    host_elapsed = None
    first_result = None
    gather_s = None
    stats = {}
    path = {}
    if not args.isolated_only:
        # W untimed warm-up steps, run the way the timed steps run (--inflight at a time; with W < inflight the warm-up batches
        # are aligned again until that many have been in flight together), so that workspaces and the host block pool reach
        # their working sizes before the clock starts
        from concurrent.futures import ThreadPoolExecutor
        if args.warmup > 0:
            with ThreadPoolExecutor(max_workers=max(1, args.inflight)) as wpool:
                n_warm = max(args.warmup, args.inflight)
                for f in [wpool.submit(al.align_reads, resident[w % args.warmup]) for w in range(n_warm)]:
                    res = f.result()
                    D.exchange_counts(res.n_records, len(res.cigars), rank, world, device=dev)
        log("warm")
        ctx.enable_kernel_timing(True)
        ctx.reset_kernel_stats()
        ctx.path_counters(reset=True)

        def timed_region(inputs):
            """K steps over `inputs` (one per timed batch): (seconds (max over ranks), records, seconds rank 0 spent gathering, result of step 0)"""
            tpool = ThreadPoolExecutor(max_workers=max(1, args.inflight))
            g_s = 0.0 if world > 1 else None
            gathered_rows = 0
            n_rec = 0
            first = None
            barrier()
            t_start = time.perf_counter()
            # a step = the whole hot path over one batch. Batches are independent (floxer itself streams reads through a thread
            # pool without a barrier between them), so up to --inflight steps are in the context at once: while one batch's lanes
            # are in a host phase another batch's kernels keep the GPU busy. All K steps start and finish inside the timed region.
            futures = [tpool.submit(al.align_reads, inputs[s % n_timed_batches]) for s in range(args.steps)]
            for si, f in enumerate(futures):
                res = f.result()
                if si == 0:
                    first = res
                # the only exchanges between ranks: the sizes of the ranks' parts of this step (= where each part goes in the job's
                # output, which is the parts in rank order), then the step's records to rank 0 - while the next steps compute
                counts = D.exchange_counts(res.n_records, len(res.cigars), rank, world, device=dev)
                n_rec += int(counts[:, 0].sum())
                if world > 1:
                    t_g = time.perf_counter()
                    if args.gather_cigars:
                        got = D.gather_records(res.rows, res.cigars, (si * world + rank) * B, rank, world, device=dev)
                        if rank == 0:
                            gathered_rows += len(got[0])
                    else:
                        rows = res.rows.copy()
                        rows[:, 0] += (si * world + rank) * B          # global read index of this step's shard
                        table = D.gather_rows(rows, counts, rank, world, device=dev)
                        if rank == 0:
                            gathered_rows += int(table.shape[0])
                            if args.dump_gathered and si == 0:
                                np.save(args.dump_gathered, np.asarray(table.cpu() if hasattr(table, "cpu") else table))
                        del table
                    g_s += time.perf_counter() - t_g
                elif args.dump_gathered and si == 0:
                    rows = res.rows.copy()
                    rows[:, 0] += (si * world + rank) * B
                    np.save(args.dump_gathered, rows)
            barrier()
            secs = time.perf_counter() - t_start
            tpool.shutdown()
            if world > 1:
                t = torch.tensor([secs], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                secs = float(t.item())
                if rank == 0:
                    assert gathered_rows == n_rec, (gathered_rows, n_rec)
            return secs, n_rec, g_s, first

        elapsed, n_records, gather_s, first_result = timed_region(resident[first_timed:first_timed + n_timed_batches])
        log(f"timed region {elapsed:.2f} s")

        stats = ctx.kernel_stats()
        path = ctx.path_counters()
        ctx.enable_kernel_timing(False)
        # ---- what floxer's own stopwatch spans (floxer.cpp:154-179: query reading to last record): the same K steps with the reads in
        #      pageable host memory, flx_align_reads: the H2D copy, the reverse complements, the 2-bit form and the Peq planes inside the clock
        if not args.no_host_inputs_leg:
            host_elapsed, host_records, _, _ = timed_region(batches[first_timed:first_timed + n_timed_batches])
            assert host_records == n_records, (host_records, n_records)
            log(f"timed region, host inputs {host_elapsed:.2f} s")
    # the timed region's context and its resident batches leave the GPU before the one-lane pass makes its own (a repeat-rich reference
    # grows the lanes' workspaces: both contexts together ran out of HBM)
    for rr in resident:
        rr.close()
    resident = []
    ctx.close()

    # ---- isolated pass (rank 0): the first timed batch once more on ONE lane, so that no two kernels overlap and a launch's
    #      HIP-event time (on the launch stream) is its own duration. Outside the timed region; feeds "roofline".
    iso_stats = {}
    parity_failed = False
    if rank == 0 and not args.no_isolated_pass:
        os.environ["FLX_LANES"] = "1"
        ctx1 = F.context(index, device=local_rank, image=image)
        al1 = F.aligner(ctx1, p)
        rr1 = F.resident_reads(ctx1, batches[first_timed])
        al1.align_reads(rr1)                                   # warm the workspaces
        ctx1.enable_kernel_timing(True)
        ctx1.reset_kernel_stats()
        iso_result = al1.align_reads(rr1)
        if first_result is None:
            first_result = iso_result
        iso_stats = ctx1.kernel_stats()
        rr1.close()
        ctx1.close()
        log("isolated pass")

    if rank == 0:
        total_reads = B * args.steps * world
        offs = batches[first_timed][1]
        mean_len = float(offs[-1]) / max(1, len(offs) - 1)
        value = total_reads / elapsed

        # ---- CPU baseline: the oracle on a sample of the same reads; it also counts the cursor extensions the reference's walk makes
        cpu = None
        parity = None
        ext_per_read = None
        ext_source = None
        if not args.no_cpu_baseline and world == 1:            # (rank 0 at N = 1 only)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O                      # the checker, timed as the reported CPU baseline only
            ncores = usable_cores()
            offs0 = batches[first_timed][1]
            n_s = min(args.cpu_sample, B)
            sample = [batches[first_timed][0][int(offs0[i]):int(offs0[i + 1])] for i in range(n_s)]
            t0 = time.time()
            # the oracle takes the suffix array and the BWTs of the reference as data (a text has one suffix array; sorting 3.1 G
            # suffixes on the CPU would take the better part of an hour) and lays out its own index around them
            oidx = O.Index(genome, imported=(index.suffix_array_u32(), index.bwt(False), index.bwt(True)), pool=pool)
            log(f"oracle index import {time.time() - t0:.1f} s")
            ores = oidx.run(sample, O.params(error_probability=args.error_rate, interval_opt=args.interval_optimization), threads=ncores)
            log(f"oracle run {ores.seconds:.1f} s")
            ext_per_read = float(int(ores.counters[0]) + int(ores.counters[1])) / max(1, len(sample))
            # the oracle's records of the sample against the product's records of the same reads (step 0 of the timed region, or the
            # one-lane pass): the comparison the reference's own end-to-end test makes (floxer_whole_program_via_cli_test.cpp:40-94),
            # at the metric's size
            parity = compare_with_oracle(first_result, ores, len(sample)) if first_result is not None else None
            log(f"parity sample {parity}")
            ext_source = f"counted by the oracle on the cpu_baseline sample ({len(sample)} reads)"
            cpu = {"value": round(len(sample) / ores.seconds, 3), "unit": "reads/s", "cores": ncores, "kind": "port",
                   "sample": f"first {len(sample)} reads of the first timed batch against the same {args.genome / 1e9:.1f} Gb reference, oracle "
                             f"(CPU restatement of floxer's path) on {ncores} threads; index build excluded on both sides (the oracle's "
                             "index is laid out around the suffix array and BWTs imported from the product's index)",
                   "cursor_extensions_per_read": round(ext_per_read, 1)}
        if ext_per_read is None:
            # no oracle run in this invocation: the count of a committed run on this workload (a function of reference, reads and flags)
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_oracle_extensions.json")))
                key = f"{args.genome}/{args.chromosomes}/{args.read_length}/{args.error_rate}/{int(args.repeat_rich)}"
                if key in t:
                    ext_per_read = float(t[key]["cursor_extensions_per_read"])
                    ext_source = f"profiles/{PROFILE_TAG}_oracle_extensions.json (oracle count of a committed run on this workload)"
            except (OSError, ValueError, KeyError):
                pass

        def load_traffic(name, st):
            """HBM bytes per launch = FETCH_SIZE + WRITE_SIZE of the committed PMC passes over this workload per work unit of the
            kernel (profiles/<tag>_pmc_traffic_<kernel>.json) x the work units of a launch here; null when no pass over this
            workload is committed"""
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_traffic_{name}.json")))
                if t.get("kernel") == name and t.get("read_length") == args.read_length and t.get("genome") == args.genome:
                    return int(st["work_units"] / st["launches"] * t["traffic_bytes_per_work_unit"])
            except (OSError, ValueError, KeyError, ZeroDivisionError):
                pass
            return None

        def roof(name, st, note, reads_per_launch=None):
            # algorithmic bytes = what THIS build's kernels must touch (flx_pipeline.cpp, k1_bytes: rank blocks, filter words, text / SA /
            # ISA, seed records, hits and queued subtrees, each from a device counter, random accesses at the 64-B request size): a
            # fraction of what the memory system can deliver, <= 1 by construction, the same model for every --config
            alg = st["algorithmic_bytes"]
            achieved = alg / 1e9 / (st["device_ms"] / 1e3)
            r = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_traffic(name, st),
                 "avg_launch_ms": round(st["device_ms"] / st["launches"], 4),
                 "algorithmic_bytes_per_launch": int(alg / st["launches"]), "launches": st["launches"],
                 "work_units_per_launch": int(st["work_units"] / st["launches"]), "note": note}
            if name in ("ed_align_trace", "ed_align_exists"):
                # The DP kernels are bit-vector arithmetic: their bound is vector-instruction issue, not HBM. One word-step (64 rows of one
                # column) is ~40 VALU instructions of one lane (flx_device.hip, ed_block_body); a wave64 instruction takes 2 cycles on a
                # CDNA4 SIMD-32, 4 when the wave is alone on its SIMD (MI355X_MICROARCH.md): 1024 SIMDs x 2.4 GHz / 2.
                wave_instr = st["work_units"] * 40.0 / 64.0
                peak = 1024 * 2.4e9 / 2
                r["valu_issue"] = {"model": "40 VALU instructions per word-step, 64 lanes per wave instruction", "wave_instructions_per_launch": int(wave_instr / st["launches"]),
                                   "achieved_G_per_s": round(wave_instr / (st["device_ms"] / 1e3) / 1e9, 1), "peak_G_per_s": round(peak / 1e9, 1),
                                   "frac": round(wave_instr / (st["device_ms"] / 1e3) / peak, 4)}
            if name == "fm_search" and ext_per_read is not None and reads_per_launch:
                # SURVEY.md 8(d) prices the seeding at 2 x 64 B per cursor extension OF THE REFERENCE'S WALK on this input (counted by the
                # restatement). This build answers the same questions with fewer rank queries (presence filter, text walk), so that
                # figure is a speed-up over a rank-query walk at the same memory rate - NOT a fraction of the HBM peak (it can exceed it)
                ref_bytes = 128.0 * ext_per_read * reads_per_launch * st["launches"]
                r["reference_walk_equivalent_GBps"] = round(ref_bytes / 1e9 / (st["device_ms"] / 1e3), 2)
                r["reference_walk_extensions_per_read"] = round(ext_per_read, 1)
            return r

        def table(sts):
            return {k: {"launches": v["launches"], "device_ms": round(v["device_ms"], 3),
                        "algorithmic_GB": round(v["algorithmic_bytes"] / 1e9, 4), "work_units": v["work_units"],
                        "GBps": round(v["algorithmic_bytes"] / 1e6 / v["device_ms"], 2) if v["device_ms"] > 0 else None}
                    for k, v in sts.items()}

        fm_note = ("fm_search = the filter walk + the text walk (two kernels, one HIP-event bracket); bytes = what this walk touches, from its "
                   "device counters: 2 x 64 B per rank pair, 64 B per filter word, per queued subtree its record written and read + SA + "
                   "text + seed record, per hit its record + ISA, per seed its record + symbols"
                   + (f"; reference_walk_equivalent_GBps prices the reference's walk instead (128 B x extensions {ext_source})" if ext_per_read is not None else ""))
        # the dominant kernel = largest device time when nothing overlaps (the one-lane pass); its roofline is that pass's:
        # summed event times of the timed region count the time a launch shares the chip with the other lanes' kernels
        roofline = roofline_timed = roofline_fm = None
        dom = None
        if iso_stats:
            dom = max(iso_stats.items(), key=lambda kv: kv[1]["device_ms"])[0]
            iso_note = "one-lane pass over the first timed batch, outside the timed region: launches do not overlap, HIP events on the launch stream; "
            roofline = roof(dom, iso_stats[dom], iso_note + (fm_note if dom == "fm_search" else "bytes = sequence bytes read + trace / last-row bytes written "
                            "(SURVEY.md 8d): a bit-vector DP kernel, VALU-bound by design, so its HBM fraction is low"),
                            reads_per_launch=B / max(1, iso_stats[dom]["launches"]))
            if "fm_search" in iso_stats:
                roofline_fm = roof("fm_search", iso_stats["fm_search"], iso_note + fm_note, reads_per_launch=B / max(1, iso_stats["fm_search"]["launches"]))
        if stats:
            name = dom if dom in stats else max(stats.items(), key=lambda kv: kv[1]["device_ms"])[0]
            roofline_timed = roof(name, stats[name], f"timed region, {args.lanes} lanes: kernels of different lanes overlap on the GPU, so a "
                                  "launch's HIP-event time includes the time it shares the chip (not a duration of its own)",
                                  reads_per_launch=B * args.steps / max(1, stats[name]["launches"]))
        if roofline is None:
            roofline = roofline_timed

        # ---- the metric says "vs hg38": half of a human genome is repeats, and there the seeding is a different workload (intervals of
        #      many rows inside diverged families, nothing for the presence filter to reject). The default run of the metric's
        #      configuration therefore carries a second, shorter measurement on the repeat-rich synthetic reference (same size, same
        #      reads, same flags), made by a child process once this one has given the GPU's memory back.
        repeat_rich_line = None
        index_device_bytes = int(index.device_bytes)
        index_derived_bytes = int(index.derived_device_bytes)
        if (world == 1 and not args.no_repeat_rich_leg and not args.repeat_rich and not args.isolated_only and args.config == "grch38"
                and (args.genome, args.read_length, args.error_rate) == (CONFIGS["grch38"]["genome"], 10000, 0.08)):
            del image, index, genome, pool, batches
            if not args.no_cpu_baseline:
                del oidx, ores
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            cmd = [sys.executable, os.path.abspath(__file__), "--repeat-rich", "--steps", str(args.repeat_rich_steps), "--warmup", "3", "--no-cpu-baseline",
                   "--no-isolated-pass", "--no-host-inputs-leg", "--lanes", str(args.lanes), "--inflight", str(args.inflight)]
            if args.interval_optimization:
                cmd.append("--interval-optimization")
            t0 = time.time()
            try:
                child = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                d = json.loads(child.stdout.strip().splitlines()[-1])
                repeat_rich_line = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "gbases_per_s": d["gbases_per_s"],
                                    "records": d["records"], "workload": d["config"]["workload"], "path": d["path"],
                                    "kernels": {k: v["device_ms"] for k, v in d["kernels"].items()}, "wall_s": round(time.time() - t0, 1)}
            except Exception as ex:          # the secondary line must not cost the primary one
                repeat_rich_line = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            log(f"repeat-rich leg {time.time() - t0:.1f} s")

        cfg = CONFIGS[args.config]
        is_cfg = (args.genome, args.chromosomes, args.read_length, args.error_rate) == (cfg["genome"], cfg["chromosomes"], cfg["read_length"], cfg["error_rate"])
        ref_kind = "repeat-rich synthetic reference (interspersed families, tandem repeats, low complexity, N runs, segmental duplications: " \
                   "half of the bases unique)" if args.repeat_rich else "uniform random reference"
        metric_cfg = args.config == "grch38" and is_cfg and not args.repeat_rich
        line = {
            "metric": f"aligned reads/sec, {args.read_length // 1000} kb reads @ {args.error_rate:.0%} error vs a {args.genome / 1e9:.2g} Gb reference "
                      "(seed-and-verify path, CIGAR)" if not metric_cfg else
                      "aligned reads/sec, 10 kb ONT-like reads @ 8 % error vs a GRCh38-size reference (seed-and-verify path, CIGAR)",
            "value": round(value, 2), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.genome / 1e9:.2g} Gb {ref_kind} in {args.chromosomes} sequence(s)"
                                   + (" (GRCh38 size; hg38 itself is not available offline)" if args.genome >= 3_000_000_000 else "")
                                   + f" + {B} reads/GPU/step of {args.read_length} bp @ {args.error_rate:.0%} error"
                                   + (f" ({cfg['name']})" if is_cfg else " (not a BASELINE.json configuration)"),
                       "config": args.config, "repeat_rich": bool(args.repeat_rich),
                       "genome": args.genome, "reads_per_step_per_gpu": B, "mean_read_length": round(mean_len, 1), "cli_flags": "defaults (-s 2 -M 500 -m 50 "
                       "-g count_first -y round_robin -v 0.05)" + (" -I" if args.interval_optimization else ""),
                       "lanes_per_gpu": args.lanes, "cores_per_rank": cores, "steps_in_flight": args.inflight, "timed_batches": n_timed_batches,
                       "parallelism": f"read-sharded x{world}, index replicated",
                       "gather": None if world == 1 else ("records + CIGAR words" if args.gather_cigars else "fixed-size records") + " to rank 0 per step, inside the timed region"},
            "gbases_per_s": round(value * mean_len / 1e9, 5), "records": n_records, "index_build_s": round(index_s, 2),
            "index_device_bytes": index_device_bytes, "index_derived_device_bytes": index_derived_bytes,
            "gather_s": None if gather_s is None else round(gather_s, 3),      # time rank 0 spent in the per-step gathers (inside the timed region)
            "roofline": roofline, "roofline_fm_search": roofline_fm, "roofline_timed_region": roofline_timed, "cpu_baseline": cpu,
            "parity_sample": parity,
            # `value` = inputs resident in HBM when the clock starts (the contract's definition); value_host_inputs = the same K steps with
            # the reads in pageable host memory (flx_align_reads: H2D, reverse complements, 2-bit form and Peq planes inside the clock) -
            # what floxer's own stopwatch spans (floxer.cpp:154-179) minus file parsing
            "value_host_inputs": None if host_elapsed is None else round(total_reads / host_elapsed, 2),
            "ms_per_step_host_inputs": None if host_elapsed is None else round(host_elapsed / args.steps * 1e3, 3),
            "repeat_rich": repeat_rich_line,
            "path": {k: int(v) for k, v in path.items()} if path else None,
            "kernels": table(stats), "kernels_isolated": table(iso_stats),
        }
        print(json.dumps(line), flush=True)
        if parity is not None and not parity["equal"]:
            print(f"bench.py: the product's records differ from the oracle's on the parity sample: {parity}", file=sys.stderr, flush=True)
            parity_failed = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
