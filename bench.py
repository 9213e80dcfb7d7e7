#!/usr/bin/env python3
"""bench.py — reads/s of the seed-and-verify hot path on MI355X.

One "step" = one pass of the whole path (PEX seeding -> FM search -> hierarchical verification -> root alignment with
CIGAR -> records) over one batch of synthetic long reads that is already resident in HBM. Reads shard across ranks with
no data-path collective (FM index replicated per GPU); every rank keeps its part of the output (the job's output is the parts
in rank order), the ranks exchange the sizes of their parts (RCCL all-gather) at the end of every step and the fixed-size
alignment records of the whole job are gathered to rank 0 once at the end (RCCL gather, inside the timed region). Prints ONE
JSON line on rank 0.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-step", type=int, default=int(os.environ.get("FLX_BENCH_READS", 16384)), help="per GPU")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("FLX_LANES", 0)),
                    help="concurrent lanes (stream + host thread) per GPU; 0 = one per host core this rank can use, 4..16")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("FLX_BENCH_INFLIGHT", 3)),
                    help="steps submitted to the context at a time (host threads calling align_reads); every step still runs in "
                         "full inside the timed region")
    ap.add_argument("--no-isolated-pass", action="store_true", help="skip the one-lane instrumented pass (timeline profiling)")
    ap.add_argument("--isolated-only", action="store_true",
                    help="only the one-lane instrumented pass (used under rocprofv3 so that its per-kernel averages are those of roofline_isolated)")
    ap.add_argument("--genome", type=int, default=4_600_000, help="synthetic reference length (E. coli K-12 size)")
    ap.add_argument("--read-length", type=int, default=5000)
    ap.add_argument("--error-rate", type=float, default=0.08)
    ap.add_argument("--interval-optimization", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("FLX_BENCH_CPU_SAMPLE", 256)))
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    if args.lanes <= 0:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        args.lanes = max(4, min(16, usable_cores() // max(1, local_world)))

    import torch
    import torch.distributed as dist
    import floxer_amd as F
    from floxer_amd import simulate as S

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- workload: BASELINE.json configs[1] shape (4.6 Mb reference, 5 kb reads @ 8 %), synthetic (no network for E. coli)
    genome = S.make_genome(args.genome, 1, seed=S.DEFAULT_SEED)
    n_batches = args.steps + args.warmup
    B = args.reads_per_step
    batches = []
    for b in range(n_batches):
        # every rank and every step gets its own reads (weak scaling: per-GPU work is fixed)
        reads, _, _ = S.make_reads(genome, B, args.read_length, args.error_rate, seed=S.DEFAULT_SEED + 1 + rank * 1000 + b)
        batches.append(reads)

    os.environ["FLX_LANES"] = str(args.lanes)
    t0 = time.time()
    index = F.fmindex(genome, device=local_rank)    # suffix arrays on the GPU; not timed (floxer's stopwatch excludes it too, floxer.cpp:154)
    index_s = time.time() - t0
    ctx = F.context(index, device=local_rank)
    p = F.params(error_probability=args.error_rate, interval_optimization=args.interval_optimization)
    al = F.aligner(ctx, p)
    resident = [F.resident_reads(ctx, r) for r in batches]       # inputs resident in HBM before the timed region

    from floxer_amd import distributed as D
    dev = torch.device("cuda", local_rank)

    def exchange(res):
        """the only exchange between ranks: the sizes of the ranks' parts (= where each part goes in the job's output, which is the
        parts in rank order); every rank keeps its own records. Returns the job's record count of this step."""
        counts = D.exchange_counts(res.n_records, len(res.cigars), rank, world, device=dev)
        return int(counts[:, 0].sum())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_records = 0
    elapsed = 1.0
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
        ctx.enable_kernel_timing(True)
        ctx.reset_kernel_stats()

        pool = ThreadPoolExecutor(max_workers=max(1, args.inflight))

        barrier()
        import resource
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        if os.environ.get("FLX_ALLOC_DEBUG"):
            print(f"[bench] {time.time():.3f} timed region starts", file=sys.stderr, flush=True)
        t_start = time.perf_counter()
        # a step = the whole hot path over one batch. Batches are independent (floxer itself streams reads through a thread
        # pool without a barrier between them), so up to --inflight steps are in the context at once: while one batch's lanes
        # are in a host phase another batch's kernels keep the GPU busy. All K steps start and finish inside the timed region.
        futures = [pool.submit(al.align_reads, resident[args.warmup + s]) for s in range(args.steps)]
        kept_rows, my_rows = [], 0
        for si, f in enumerate(futures):
            res = f.result()
            n_records += exchange(res)
            if world > 1:
                rows = res.rows.copy()
                rows[:, 0] += (si * world + rank) * B          # global read index of this step's shard
                kept_rows.append(rows)
                my_rows += len(rows)
        if world > 1:
            # the job's one gather (RCCL over xGMI): the fixed-size alignment records of all K steps go to rank 0 and stay in its
            # HBM; CIGAR words stay in the owners' parts (see floxer_amd/distributed.py)
            totals = D.exchange_counts(my_rows, 0, rank, world, device=dev)
            table = D.gather_rows(np.concatenate(kept_rows, axis=0) if kept_rows else np.zeros((0, 7), np.int64), totals, rank, world, device=dev)
            if rank == 0:
                assert sum(int(t.shape[0]) for t in table) == n_records
        barrier()
        elapsed = time.perf_counter() - t_start
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        if os.environ.get("FLX_ALLOC_DEBUG"):
            print(f"[bench] {time.time():.3f} timed region ends", file=sys.stderr, flush=True)
            print(f"[bench] timed region: minor page faults {ru1.ru_minflt - ru0.ru_minflt}, user {ru1.ru_utime - ru0.ru_utime:.3f} s, "
                  f"system {ru1.ru_stime - ru0.ru_stime:.3f} s, wall {elapsed:.3f} s", file=sys.stderr, flush=True)
        pool.shutdown()
        if world > 1:
            t = torch.tensor([elapsed], device=torch.device("cuda", local_rank), dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        stats = ctx.kernel_stats()
        ctx.enable_kernel_timing(False)

    # ---- isolated pass (rank 0): the first timed batch once more on ONE lane, so that no two kernels overlap and a kernel's
    #      HIP-event time is its own duration. Outside the timed region; feeds "roofline_isolated".
    iso_stats = {}
    if rank == 0 and not args.no_isolated_pass:
        os.environ["FLX_LANES"] = "1"
        ctx1 = F.context(index, device=local_rank)
        al1 = F.aligner(ctx1, p)
        rr1 = F.resident_reads(ctx1, batches[args.warmup])
        al1.align_reads(rr1)                                   # warm the workspaces
        ctx1.enable_kernel_timing(True)
        ctx1.reset_kernel_stats()
        al1.align_reads(rr1)
        iso_stats = ctx1.kernel_stats()
        rr1.close()
        ctx1.close()

    if rank == 0:
        total_reads = B * args.steps * world
        mean_len = float(np.mean([len(r) for r in batches[args.warmup]]))
        value = total_reads / elapsed
        # ---- roofline of the dominant kernel (largest summed device time in the timed region, HIP events on the launch stream)
        # the dominant kernel is chosen on the isolated pass (un-overlapped durations); summed event times of the timed region
        # count the time a launch shares the chip with the other lanes' kernels
        if iso_stats:
            dom_name = max(iso_stats.items(), key=lambda kv: kv[1]["device_ms"])[0]
            dom = (dom_name, stats[dom_name]) if dom_name in stats else None
        else:
            dom = max(stats.items(), key=lambda kv: kv[1]["device_ms"]) if stats else None
        roofline = None
        kernels = {}
        for name, st in stats.items():
            ms = st["device_ms"]
            kernels[name] = {"launches": st["launches"], "device_ms": round(ms, 3),
                             "algorithmic_GB": round(st["algorithmic_bytes"] / 1e9, 4), "work_units": st["work_units"],
                             "GBps": round(st["algorithmic_bytes"] / 1e6 / ms, 2) if ms > 0 else None}
        def roof(name, st, note):
            achieved = st["algorithmic_bytes"] / 1e9 / (st["device_ms"] / 1e3)
            return {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_traffic(name, st),
                    "avg_launch_ms": round(st["device_ms"] / st["launches"], 4),
                    "algorithmic_bytes_per_launch": int(st["algorithmic_bytes"] / st["launches"]), "launches": st["launches"],
                    # the path is integer bit-vector work: what binds these kernels is VALU issue, not HBM and not MFMA. The
                    # fraction of VALU issue slots in use comes from the committed SQ counter pass over this workload.
                    "valu_busy_frac": load_valu(name), "note": note}

        def load_valu(name):
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_valu.json")))
                return t["kernels"][name]["valu_busy_frac"]
            except (OSError, ValueError, KeyError):
                return None

        def load_traffic(name, st):
            """HBM bytes per launch = algorithmic bytes per launch x (PMC bytes / algorithmic bytes) of the committed FETCH_SIZE /
            WRITE_SIZE passes over this workload (profiles/r01_pmc_traffic_<kernel>.json; launches per pass vary with the trace
            arena, the ratio does not)"""
            try:
                t = json.load(open(os.path.join(ROOT, "profiles", f"r01_pmc_traffic_{name}.json")))
                if t.get("kernel") == name and t.get("read_length") == args.read_length:
                    return int(st["algorithmic_bytes"] / st["launches"] * t["traffic_over_algorithmic"])
            except (OSError, ValueError, KeyError):
                pass
            return None

        roofline_iso = None
        if iso_stats:
            iname = max(iso_stats.items(), key=lambda kv: kv[1]["device_ms"])[0]
            roofline_iso = roof(iname, iso_stats[iname], "one-lane pass of the first timed batch outside the timed region: kernels do not overlap")
            if not dom:
                dom = (iname, iso_stats[iname])
        if dom and stats:
            roofline = roof(dom[0], dom[1], f"timed region, {args.lanes} lanes: kernels of different lanes overlap on the GPU, so a launch's "
                                             "HIP-event time includes the time it shares the chip")
        else:
            roofline = roofline_iso

        cpu = None
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O                      # the checker, timed as the reported CPU baseline only
            cores = usable_cores()
            sample = batches[args.warmup][: args.cpu_sample]
            oidx = O.Index(genome)
            ores = oidx.run(sample, O.params(error_probability=args.error_rate, interval_opt=args.interval_optimization), threads=cores)
            cpu = {"value": round(len(sample) / ores.seconds, 3), "unit": "reads/s", "cores": cores, "kind": "port",
                   "sample": f"first {len(sample)} reads of the first timed batch, oracle (CPU restatement of floxer's path), "
                             f"{cores} threads, index build excluded"}
        line = {
            "metric": "aligned reads/sec, simulated long reads vs synthetic reference (seed-and-verify path, CIGAR)",
            "value": round(value, 2), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.genome / 1e6:.1f} Mb uniform random reference + {B} reads/GPU/step of {args.read_length} bp "
                                   f"@ {args.error_rate:.0%} error (BASELINE.json configs[1] shape; E. coli itself is not available offline)",
                       "reads_per_step_per_gpu": B, "mean_read_length": round(mean_len, 1), "cli_flags": "defaults (-s 2 -M 500 -m 50 "
                       "-g count_first -y round_robin -v 0.05)" + (" -I" if args.interval_optimization else ""),
                       "lanes_per_gpu": args.lanes, "steps_in_flight": args.inflight, "parallelism": f"read-sharded x{world}, index replicated"},
            "gbases_per_s": round(value * mean_len / 1e9, 5), "records": n_records, "index_build_s": round(index_s, 2),
            "roofline": roofline, "roofline_isolated": roofline_iso, "cpu_baseline": cpu, "kernels": kernels,
            "kernels_isolated": {k: {"launches": v["launches"], "device_ms": round(v["device_ms"], 3),
                                     "GBps": round(v["algorithmic_bytes"] / 1e6 / v["device_ms"], 2) if v["device_ms"] > 0 else None}
                                 for k, v in iso_stats.items()},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
