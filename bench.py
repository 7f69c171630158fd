#!/usr/bin/env python3
"""Benchmark of the compare hot path on MI355X.

One step = one pass of the per-region solver (phasing search + genotype assignment + metrics,
reference solve_compare_region, src/waffle_solver.rs:122) over one resident batch: BASELINE.json
configs[1] "Synthetic chr20: 50k SNV-only truth vs query, confident BED" (~48k regions).
Inputs (reference genome, region batch) are resident in HBM before the timed region starts.
With N > 1 ranks every rank owns its own chr20-sized call set (weak scaling: independent
confident-region blocks are sharded, no data-path collective); every step adds its per-category
tallies to the job total on the device, and the job total is summed over the ranks with one RCCL
all-reduce at the end of the timed region.

Prints ONE JSON line on rank 0 (see the contract in the task statement).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n-truth", type=int, default=50_000, help="truth SNVs of the synthetic chr20 call set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the bit-identity gate against the oracle")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"),
                    help="PMC-derived HBM bytes per launch collected with rocprofv3 --pmc (optional)")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # direct invocation: start the launcher as a child before anything touches the GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # stdout carries exactly ONE line, the JSON of rank 0: everything else that writes to file descriptor 1 — RCCL prints a version
    # banner there when a communicator comes up — is sent to stderr; the JSON goes to the original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import aardvark_amd
    from aardvark_amd import CompareConfig, synth

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("AVK_BENCH_FORCE_DIST") == "1"  # the second form exercises the collective path on one GPU
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    # ---- workload: one synthetic chr20 call-set pair per rank (seeds of SURVEY.md §8d, offset by rank)
    contig, batch = synth.config_chr20_snv(n_truth=args.n_truth, seed_ref=20250101 + 1000 * rank, seed_query=20250102 + 1000 * rank)
    n_regions = batch.n_regions

    ctx = aardvark_amd.Context(local_rank)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)  # launches go on torch's current stream
    ctx.set_option("emit_group_metrics", 0)  # per-variant decisions + batch tally only
    for kv in os.environ.get("AVK_OPTS", "").split(","):  # tuning experiments: context options by name
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.upload_reference([contig])
    rb = ctx.upload(batch)
    cfg = CompareConfig(enable_sequences=False)
    # The job's tally: every step (= one batch of the job) adds its tally block to a running total on the device
    # (SummaryWriter::add_comparison_benchmark, writers/summary.rs:146-163); the total is summed over the ranks with ONE
    # RCCL all-reduce when the job's batches are done — inside the timed region.  There is no data-path collective.
    ctx.set_option("accumulate_tally", 1)
    tally = torch.zeros(aardvark_amd.TALLY_LEN, dtype=torch.int64, device=dev)

    def step():
        ctx.compare_resident(rb, cfg, tally.data_ptr())

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.all_reduce(tally, op=dist.ReduceOp.SUM)  # warm the communicator up as well
    fence()
    tally.zero_()
    fence()
    kernel_ms, solver_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(ctx.last_kernel_ms())  # hipEvents around the dominant launch, recorded on the launch stream
        solver_ms.append(ctx.last_solver_ms())
    if use_dist:
        dist.all_reduce(tally, op=dist.ReduceOp.SUM)  # RCCL over xGMI: 288 x int64
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([n_regions], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_regions = int(cnt.item())
    else:
        total_regions = n_regions

    # ---- bit-identity gate (same-run rule): this rank's outputs against the oracle
    parity = None
    if not args.no_parity:
        import oracle_lib
        lib = oracle_lib.load()
        got = ctx.download(rb, group_metrics=False)
        want = oracle_lib.compare_batch(lib, batch, [contig], threads=min(os.cpu_count() or 1, 64))
        want.group_metrics = None
        bad = got.diff(want)
        # the job total must be steps x this rank's tally, summed over the ranks
        mine = torch.from_numpy(want.tally.astype(np.int64)).to(dev) * args.steps
        if use_dist:
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        if not torch.equal(mine, tally):
            bad.append("job_tally")
        parity = "bit-identical" if not bad else "MISMATCH:" + ",".join(bad)
        ok = torch.tensor([0 if bad else 1], device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            print("PARITY FAILURE on rank %d: %s" % (rank, parity), file=sys.stderr)
            sys.exit(3)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_regions * args.steps / elapsed
        alg_bytes = ctx.algorithmic_bytes(batch)
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                traffic = json.load(open(args.traffic_json)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "compared regions/sec (whole node)",
            "value": value,
            "unit": "regions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "synthetic chr20 (64,444,167 bp): %d SNV-only truth vs query calls, confident BED, %d regions per GPU"
                                   % (args.n_truth, n_regions),
                       "regions_per_gpu": n_regions, "max_branch_factor": cfg.max_branch_factor, "min_variant_gap": 50,
                       "parallelism": "regions sharded over %d GPU(s), no data-path collective; one RCCL all-reduce of the job tally (288 x int64) inside the timed region" % world,
                       "parity": parity},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "avk_region_kernel_lds (first pass)", "kernel_ms": k_ms, "all_solver_launches_ms": float(np.mean(solver_ms)), "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_region": alg_bytes / max(n_regions, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib
            lib = oracle_lib.load()
            cores = os.cpu_count() or 1
            reps = max(1, int(15.0 * 70_000 / max(n_regions, 1)))  # ~15 core-seconds of oracle work
            oracle_lib.bench(lib, batch.slice(0, min(n_regions, 4096)), [contig], cores, 1)  # spin the threads up once
            sec, rate = oracle_lib.bench(lib, batch, [contig], cores, reps)
            sec1, rate1 = oracle_lib.bench(lib, batch, [contig], 1, 1)
            out["cpu_baseline"] = {"value": rate, "unit": "regions/s", "cores": cores, "kind": "port",
                                   "sample": "%d passes over the same %d-region batch (%.2f s wall on %d threads); 1 thread: %.0f regions/s"
                                             % (reps, n_regions, sec, cores, rate1)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
