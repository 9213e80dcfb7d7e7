#!/usr/bin/env python3
"""bench.py — reads/s of the seed-and-verify hot path on MI355X, on the configuration BASELINE.json's metric is quoted on:
10 kb reads @ 8 % error against a GRCh38-size reference (3.1 Gb in 25 sequences; synthetic, uniform over ACGT: hg38 itself is not
available offline), floxer's default flags.

One "step" = one pass of the whole path (PEX seeding -> FM search -> hierarchical verification -> root alignment with CIGAR ->
records) over one batch of synthetic long reads that is already resident in HBM. Reads shard across ranks with no data-path
collective (FM index replicated per GPU); every rank keeps its part of the output (the job's output is the parts in rank order),
the ranks exchange the sizes of their parts (RCCL all-gather) at the end of every step and the fixed-size alignment records of the
whole job are gathered to rank 0 once after the clock has stopped (RCCL gatherv over xGMI; its time is reported as "gather_s"). Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The lanes of a context are HIP streams; ROCm multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and kernels of streams that share a queue do not overlap. Read by the HIP runtime when it initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_TAG = "r02"        # profiles/<tag>_pmc_traffic_<kernel>.json: committed FETCH_SIZE / WRITE_SIZE passes over this workload


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-step", type=int, default=int(os.environ.get("FLX_BENCH_READS", 16384)), help="per GPU")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("FLX_LANES", 0)),
                    help="concurrent lanes (stream + host thread) per GPU; 0 = 16")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("FLX_BENCH_INFLIGHT", 3)),
                    help="steps submitted to the context at a time (host threads calling align_reads); every step still runs in "
                         "full inside the timed region")
    ap.add_argument("--no-isolated-pass", action="store_true", help="skip the one-lane instrumented pass (timeline profiling)")
    ap.add_argument("--isolated-only", action="store_true",
                    help="only the one-lane instrumented pass (used under rocprofv3 so that its per-kernel averages are those of roofline)")
    ap.add_argument("--genome", type=int, default=3_100_000_000, help="synthetic reference length in total (GRCh38 size)")
    ap.add_argument("--chromosomes", type=int, default=25, help="sequences the reference is cut into (hg38: 22 + X + Y + M)")
    ap.add_argument("--read-length", type=int, default=10000)
    ap.add_argument("--error-rate", type=float, default=0.08)
    ap.add_argument("--interval-optimization", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("FLX_BENCH_CPU_SAMPLE", 768)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


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


def log(msg):
    if os.environ.get("FLX_BENCH_VERBOSE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    cores = max(1, usable_cores() // max(1, local_world))
    os.environ.setdefault("FLX_SIM_THREADS", str(cores))
    if args.lanes <= 0:
        # A lane is a stream and a host thread that sleeps while its chunk is on the GPU: the number of lanes is what the GPU needs to
        # have chunks in every stage (16), not the number of cores (measured with 16 lanes: 16 cores 81 k reads/s, 4 cores 77.6 k,
        # 3 cores 73.8 k, 2 cores 66.5 k; 8 lanes on 16 cores: 72.6 k)
        args.lanes = 16

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

    # ---- workload: BASELINE.json's metric configuration (configs[3] per GPU): GRCh38-size reference, 10 kb reads @ 8 %
    t0 = time.time()
    chrom_len = args.genome // args.chromosomes
    pool, genome = S.make_genome_fast(chrom_len, args.chromosomes, seed=S.DEFAULT_SEED)
    chrom_lens = [chrom_len] * args.chromosomes
    log(f"genome {time.time() - t0:.1f} s")
    # Every step has its own batch of reads, resident in HBM before the clock starts (0.57 GB per batch with its Peq planes) - up
    # to FLX_BENCH_MAX_BATCHES (48) of them: a longer run goes through the timed batches again in turn (nothing of an earlier pass
    # over a batch is kept but its Peq planes, 0.25 ms of kernel time per step), so that --steps 200 does not ask for 120 GB of reads
    # next to the index and the lanes' workspaces.
    n_timed_batches = min(args.steps, int(os.environ.get("FLX_BENCH_MAX_BATCHES", 48)))
    n_batches = n_timed_batches + args.warmup
    B = args.reads_per_step
    t0 = time.time()
    # every rank and every step gets its own reads (weak scaling: per-GPU work is fixed)
    batches = [S.make_reads_fast(pool, chrom_lens, B, args.read_length, args.error_rate, seed=S.DEFAULT_SEED + 1 + rank * 1000 + b)[0]
               for b in range(n_batches)]
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

    def exchange(res):
        """the only exchange between ranks per step: the sizes of the ranks' parts (= where each part goes in the job's output,
        which is the parts in rank order); every rank keeps its own records. Returns the job's record count of this step."""
        counts = D.exchange_counts(res.n_records, len(res.cigars), rank, world, device=dev)
        return int(counts[:, 0].sum())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_records = 0
    elapsed = 1.0
    gather_s = None
    stats = {}
    if not args.isolated_only:
        # W untimed warm-up steps, run the way the timed steps run (--inflight at a time; with W < inflight the warm-up batches
        # are aligned again until that many have been in flight together), so that workspaces and the host block pool reach
        # their working sizes before the clock starts
        from concurrent.futures import ThreadPoolExecutor
        if args.warmup > 0:
            with ThreadPoolExecutor(max_workers=max(1, args.inflight)) as wpool:
                n_warm = max(args.warmup, args.inflight)
                for f in [wpool.submit(al.align_reads, resident[w % args.warmup]) for w in range(n_warm)]:
                    exchange(f.result())
        log("warm")
        ctx.enable_kernel_timing(True)
        ctx.reset_kernel_stats()

        tpool = ThreadPoolExecutor(max_workers=max(1, args.inflight))

        barrier()
        t_start = time.perf_counter()
        # a step = the whole hot path over one batch. Batches are independent (floxer itself streams reads through a thread
        # pool without a barrier between them), so up to --inflight steps are in the context at once: while one batch's lanes
        # are in a host phase another batch's kernels keep the GPU busy. All K steps start and finish inside the timed region.
        futures = [tpool.submit(al.align_reads, resident[args.warmup + s % n_timed_batches]) for s in range(args.steps)]
        results = []
        for si, f in enumerate(futures):
            res = f.result()
            n_records += exchange(res)
            if world > 1:
                results.append(res)
        barrier()
        elapsed = time.perf_counter() - t_start
        tpool.shutdown()
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # When the clock stops every rank holds the records of its shards in host memory, exactly as the single rank of N = 1 does
        # (floxer's processes write their own output files). Collecting them on rank 0 is not part of the path: it is done here,
        # after the timed region, and its time reported as "gather_s" (RCCL over xGMI, a true gatherv: floxer_amd/distributed.py).
        gather_s = None
        if world > 1:
            t_g = time.perf_counter()
            kept_rows = []
            for si, res in enumerate(results):
                rows = res.rows.copy()
                rows[:, 0] += (si * world + rank) * B          # global read index of this step's shard
                kept_rows.append(rows)
            mine = np.concatenate(kept_rows, axis=0) if kept_rows else np.zeros((0, 7), np.int64)
            totals = D.exchange_counts(len(mine), 0, rank, world, device=dev)
            table = D.gather_rows(mine, totals, rank, world, device=dev)
            if rank == 0:
                assert int(table.shape[0]) == n_records
            barrier()
            gather_s = time.perf_counter() - t_g
            del table, mine, kept_rows, results
        log(f"timed region {elapsed:.2f} s")

        stats = ctx.kernel_stats()
        ctx.enable_kernel_timing(False)

    # ---- isolated pass (rank 0): the first timed batch once more on ONE lane, so that no two kernels overlap and a launch's
    #      HIP-event time (on the launch stream) is its own duration. Outside the timed region; feeds "roofline".
    iso_stats = {}
    if rank == 0 and not args.no_isolated_pass:
        os.environ["FLX_LANES"] = "1"
        ctx1 = F.context(index, device=local_rank, image=image)
        al1 = F.aligner(ctx1, p)
        rr1 = F.resident_reads(ctx1, batches[args.warmup])
        al1.align_reads(rr1)                                   # warm the workspaces
        ctx1.enable_kernel_timing(True)
        ctx1.reset_kernel_stats()
        al1.align_reads(rr1)
        iso_stats = ctx1.kernel_stats()
        rr1.close()
        ctx1.close()
        log("isolated pass")

    if rank == 0:
        total_reads = B * args.steps * world
        offs = batches[args.warmup][1]
        mean_len = float(offs[-1]) / max(1, len(offs) - 1)
        value = total_reads / elapsed

        def load_traffic(name, st):
            """HBM bytes per launch = algorithmic bytes per launch x (PMC bytes / algorithmic bytes) of the committed FETCH_SIZE /
            WRITE_SIZE passes over this workload (profiles/<tag>_pmc_traffic_<kernel>.json; launches per pass vary with the
            chunking, the ratio does not); null when no pass over this workload is committed"""
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_traffic_{name}.json")))
                if t.get("kernel") == name and t.get("read_length") == args.read_length and t.get("genome") == args.genome:
                    return int(st["algorithmic_bytes"] / st["launches"] * t["traffic_over_algorithmic"])
            except (OSError, ValueError, KeyError):
                pass
            return None

        def roof(name, st, note):
            achieved = st["algorithmic_bytes"] / 1e9 / (st["device_ms"] / 1e3)
            return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_traffic(name, st),
                    "avg_launch_ms": round(st["device_ms"] / st["launches"], 4),
                    "algorithmic_bytes_per_launch": int(st["algorithmic_bytes"] / st["launches"]), "launches": st["launches"],
                    "work_units_per_launch": int(st["work_units"] / st["launches"]), "note": note}

        def table(sts):
            return {k: {"launches": v["launches"], "device_ms": round(v["device_ms"], 3),
                        "algorithmic_GB": round(v["algorithmic_bytes"] / 1e9, 4), "work_units": v["work_units"],
                        "GBps": round(v["algorithmic_bytes"] / 1e6 / v["device_ms"], 2) if v["device_ms"] > 0 else None}
                    for k, v in sts.items()}

        # the dominant kernel = largest device time when nothing overlaps (the one-lane pass); its roofline is that pass's:
        # summed event times of the timed region count the time a launch shares the chip with the other lanes' kernels
        roofline = roofline_timed = None
        dom = None
        if iso_stats:
            dom = max(iso_stats.items(), key=lambda kv: kv[1]["device_ms"])[0]
            roofline = roof(dom, iso_stats[dom], "one-lane pass over the first timed batch, outside the timed region: launches do not overlap, "
                            "HIP events on the launch stream; fm_search bytes = 2 x 64 B per cursor extension (SURVEY.md 8d), "
                            "extensions counted by the kernel")
        if stats:
            name = dom if dom in stats else max(stats.items(), key=lambda kv: kv[1]["device_ms"])[0]
            roofline_timed = roof(name, stats[name], f"timed region, {args.lanes} lanes: kernels of different lanes overlap on the GPU, so a "
                                  "launch's HIP-event time includes the time it shares the chip (not a duration of its own)")
        if roofline is None:
            roofline = roofline_timed

        cpu = None
        if not args.no_cpu_baseline and world == 1:            # (rank 0 at N = 1 only)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O                      # the checker, timed as the reported CPU baseline only
            ncores = usable_cores()
            offs0 = batches[args.warmup][1]
            n_s = min(args.cpu_sample, B)
            sample = [batches[args.warmup][0][int(offs0[i]):int(offs0[i + 1])] for i in range(n_s)]
            t0 = time.time()
            # the oracle takes the suffix array and the BWTs of the reference as data (a text has one suffix array; sorting 3.1 G
            # suffixes on the CPU would take the better part of an hour) and lays out its own index around them
            oidx = O.Index(genome, imported=(index.suffix_array_u32(), index.bwt(False), index.bwt(True)), pool=pool)
            log(f"oracle index import {time.time() - t0:.1f} s")
            ores = oidx.run(sample, O.params(error_probability=args.error_rate, interval_opt=args.interval_optimization), threads=ncores)
            log(f"oracle run {ores.seconds:.1f} s")
            cpu = {"value": round(len(sample) / ores.seconds, 3), "unit": "reads/s", "cores": ncores, "kind": "port",
                   "sample": f"first {len(sample)} reads of the first timed batch against the same {args.genome / 1e9:.1f} Gb reference, oracle "
                             f"(CPU restatement of floxer's path) on {ncores} threads; index build excluded on both sides (the oracle's "
                             "index is laid out around the suffix array and BWTs imported from the product's index)"}
        full_size = args.genome >= 3_000_000_000 and args.read_length == 10000
        line = {
            "metric": "aligned reads/sec, 10 kb ONT-like reads @ 8 % error vs a GRCh38-size reference (seed-and-verify path, CIGAR)",
            "value": round(value, 2), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.genome / 1e9:.1f} Gb uniform random reference in {args.chromosomes} sequences "
                                   + ("(GRCh38 size; hg38 itself is not available offline)" if full_size else "(REDUCED reference: not the metric's configuration)")
                                   + f" + {B} reads/GPU/step of {args.read_length} bp @ {args.error_rate:.0%} error"
                                   + (" (BASELINE.json configs[3] shape per GPU, the metric's configuration)" if full_size else ""),
                       "genome": args.genome, "reads_per_step_per_gpu": B, "mean_read_length": round(mean_len, 1), "cli_flags": "defaults (-s 2 -M 500 -m 50 "
                       "-g count_first -y round_robin -v 0.05)" + (" -I" if args.interval_optimization else ""),
                       "lanes_per_gpu": args.lanes, "steps_in_flight": args.inflight, "parallelism": f"read-sharded x{world}, index replicated"},
            "gbases_per_s": round(value * mean_len / 1e9, 5), "records": n_records, "index_build_s": round(index_s, 2),
            "index_device_bytes": int(index.device_bytes),
            "gather_s": None if gather_s is None else round(gather_s, 3),      # records of all ranks onto rank 0, after the timed region
            "roofline": roofline, "roofline_timed_region": roofline_timed, "cpu_baseline": cpu,
            "kernels": table(stats), "kernels_isolated": table(iso_stats),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
