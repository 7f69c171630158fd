#!/usr/bin/env python3
"""Benchmark of the compare hot path on MI355X.

Workload (default): the stand-in for BASELINE.json configs[2] "HG002 GIAB v4.2.1 truth vs DeepVariant query, GRCh38, SNV+indel,
1xMI355X" — the configuration the metric is quoted on; the real GIAB files are not available offline, so the call sets are synthetic
(aardvark_amd/synth.py::config_genome): 24 contigs with GRCh38 primary lengths (3.1 Gbp), 3.9 M truth calls
(82 % SNV, 9 % insertion, 9 % deletion, 3 % of the sites within 30 bp of another, 2 % multi-allelic, 5 % of the indels in a repeat run with
the query record shifted by whole units), query = truth with 1 % dropped / 0.5 % zygosity flips / 0.5 % ALT changes / 1 % extra calls,
seeds 20250103/4: about 3.57 M regions.

One step = one pass of the per-region solver (phasing search + genotype assignment + metrics, reference solve_compare_region,
src/waffle_solver.rs:122; the rayon loop of src/main.rs:251-268) over the whole resident batch; every step adds its per-category tallies
to the job total on the device.  `value` = regions/s of the timed steps with the inputs (reference genome, region batch) resident in HBM.
Beside it, in the same JSON line:
  host_boundary   the rate of avk_compare_batch on the same batch: region batch in host memory -> packing -> H2D -> kernels -> D2H ->
                  per-region records, per-variant decisions and tally in the caller's host arrays (the boundary of the reference's loop,
                  SURVEY.md 8d);
  roofline        algorithmic bytes of the batch / duration of all solver launches of a step (HIP events on the launch stream);
  cpu_baseline    the CPU restatement of the reference algorithm (oracle/) on the usable host cores, same outputs, >= 5 s of wall time;
  dwfa_byte_compares_per_s   base comparisons the reference algorithm makes on this batch (counted by the oracle) x steps / time;
  secondary       the resident rate on configs[1] (synthetic chr20, 50 k SNV calls).

N > 1 ranks (launched by torch.distributed.run, one rank per GPU): regions are independent, so there is no data-path collective; the job
tally is summed over the ranks by ONE RCCL all-reduce inside the timed region.  `--scaling weak` (default): every rank owns a whole-genome
call set of its own (seeds offset by the rank).  `--scaling strong`: ONE call set, rank r solves the regions with
hash(region_id) % N == r (aardvark_amd/dist.py), the reference is replicated.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before anything loads the HIP runtime: the solver's six streams need queues of their own (aardvark_amd/__init__.py)
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def usable_cpus():
    """CPUs this process may use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256 logical CPUs and grant 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0, help="shrinks the contigs (and the call counts with them); 1.0 = the named workload")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--boundary-calls", type=int, default=3, help="timed avk_compare_batch calls of the host-boundary leg (0 = skip)")
    ap.add_argument("--watchdog-seconds", type=int, default=600, help="a run that takes longer dumps every thread's stack to stderr and exits (0 = off)")
    ap.add_argument("--no-supervisor", action="store_true",
                    help="run in this process (default for N > 1 ranks and under a profiler): otherwise the single-GPU run is a child of a thin supervisor "
                         "that has not touched the GPU, ends a run that exceeds --watchdog-seconds and starts it once more")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the bit-identity gate against the oracle")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r02_pmc_traffic.json"),
                    help="PMC-derived HBM bytes per step collected with rocprofv3 --pmc (optional)")
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

    # Single-GPU runs go through a supervisor: this process (standard library only, no GPU) starts the benchmark as a child, waits at most
    # --watchdog-seconds (+ 60 s for the child's own stack dump) and starts it ONE more time if it had to end it.  One stuck start was seen in
    # some eighty runs of this benchmark on fresh boxes (a first queued step that never finished; not reproduced in 30,000 steps since).
    profiled = any(k in os.environ for k in ("ROCPROFILER_REGISTER_LIBRARY", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.no_supervisor and not profiled and os.environ.get("AVK_BENCH_CHILD") != "1":
        env = dict(os.environ, AVK_BENCH_CHILD="1")
        rc = 1
        for attempt in range(2):
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True)
            try:
                rc = child.wait(timeout=args.watchdog_seconds + 60 if args.watchdog_seconds > 0 else None)
                break
            except subprocess.TimeoutExpired:
                print("[bench supervisor] attempt %d exceeded %d s: ending it%s" % (attempt + 1, args.watchdog_seconds + 60, ", starting once more" if attempt == 0 else ""),
                      file=sys.stderr, flush=True)
                try:
                    os.killpg(child.pid, 9)
                except Exception:
                    child.kill()
                child.wait()
                rc = 124
        sys.exit(rc)

    if args.watchdog_seconds > 0:  # a stuck run must end with evidence instead of holding the machine
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)

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
    from aardvark_amd import dist as avk_dist
    from aardvark_amd._abi import ResultBatch
    import ctypes as C

    def log(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_begin, msg), file=sys.stderr, flush=True)

    t_begin = time.perf_counter()
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

    # ---- workload
    cpus = usable_cpus()
    seed_shift = 1000 * rank if args.scaling == "weak" else 0
    contigs, batch = synth.config_genome(scale=args.scale, seed_ref=20250103 + seed_shift, seed_query=20250104 + seed_shift,
                                         threads=max(1, min(8, cpus // max(1, world))))
    n_job_regions = batch.n_regions
    job_batch = batch  # strong scaling: every rank holds the job's call set and solves its hash shard of it
    if args.scaling == "strong" and world > 1:
        batch = avk_dist.shard_batch(batch, rank, world)
    n_regions = batch.n_regions
    log("workload: %d contigs, %d bases, %d regions on this rank, %d calls" % (len(contigs), sum(c.size for c in contigs), n_regions, batch.n_variants))

    ctx = aardvark_amd.Context(local_rank)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)  # launches go on torch's current stream
    ctx.set_option("emit_group_metrics", 0)  # per-region records + per-variant decisions + the tally; no 1144-byte block per region
    for kv in os.environ.get("AVK_OPTS", "").split(","):  # tuning experiments: context options by name
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.upload_reference(contigs)
    rb = ctx.upload(batch)
    cfg = CompareConfig(enable_sequences=False)
    # The job's tally: every step (= one batch of the job) adds its tally block to a running total on the device
    # (SummaryWriter::add_comparison_benchmark, writers/summary.rs:146-163); the total is summed over the ranks with ONE
    # RCCL all-reduce when the job's batches are done — inside the timed region.  There is no data-path collective.
    ctx.set_option("accumulate_tally", 1)
    tally = torch.zeros(aardvark_amd.TALLY_LEN, dtype=torch.int64, device=dev)
    log("reference and batch resident")

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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
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
    log("timed region: %d steps in %.3f s" % (args.steps, elapsed))
    job_tally = tally.clone()

    # kernel durations for the roofline: HIP events the library records on the launch stream, read on steps OUTSIDE the timed region
    # (reading them synchronises the host with the step)
    ctx.set_option("accumulate_tally", 0)
    kernel_ms, solver_ms = [], []
    for _ in range(min(5, max(2, args.steps))):
        ctx.compare_resident(rb, cfg, None)
        kernel_ms.append(ctx.last_kernel_ms())
        solver_ms.append(ctx.last_solver_ms())
    tiers = lane_regions = None
    got = ctx.download(rb, group_metrics=False)
    try:
        tiers = ctx.last_tier_counts()  # regions finished per workspace tier of the wave-per-region kernels, then capacity failures
        lane_regions = ctx.last_lane_solved()  # regions finished by the lane-per-region kernel
    except Exception:
        pass

    # ---- host-boundary leg: avk_compare_batch on the same batch, host arrays in, host arrays out
    boundary = None
    if args.boundary_calls > 0:
        res = ResultBatch(batch, sequences=False, group_metrics=False)
        cb, ccfg, ro = batch.c_struct(), cfg.c_struct(), res.c_struct()
        ctx._check(ctx.lib.avk_compare_batch(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))  # untimed first call
        ctx.synchronize()
        tb = time.perf_counter()
        for _ in range(args.boundary_calls):
            ctx._check(ctx.lib.avk_compare_batch(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
        ctx.synchronize()
        b_elapsed = time.perf_counter() - tb
        same = res.diff(got) == []
        boundary = {"value": n_regions * args.boundary_calls / b_elapsed, "unit": "regions/s (this rank)", "ms_per_call": b_elapsed / args.boundary_calls * 1e3,
                    "calls": args.boundary_calls, "identical_to_resident_path": same,
                    "what": "avk_compare_batch: region batch in host memory -> validation + packing (host threads) -> H2D -> solver launches -> D2H -> "
                            "per-region records, per-variant decisions and tally in caller-owned host arrays (reference loop src/main.rs:251-268)"}
        log("host boundary: %.1f ms per call" % boundary["ms_per_call"])

    # ---- bit-identity gate (same-run rule): this rank's outputs against the oracle
    parity = None
    byte_compares = None
    cpu_rate_parity = None
    if not args.no_parity:
        import oracle_lib
        lib = oracle_lib.load()
        cs = oracle_lib.ContigSet(contigs)
        tp = time.perf_counter()
        want = oracle_lib.compare_batch(lib, batch, cs, threads=cpus, group_metrics=False)
        cpu_rate_parity = (time.perf_counter() - tp, n_regions)
        byte_compares = oracle_lib.stats(lib).get("byte_compares")
        bad = got.diff(want)
        # the job total must be steps x this rank's tally, summed over the ranks
        mine = torch.from_numpy(want.tally.astype(np.int64)).to(dev) * args.steps
        if use_dist:
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        if not torch.equal(mine, job_tally):
            bad.append("job_tally")
        if args.scaling == "strong" and world > 1:
            # the shards' per-variant decisions, gathered as ONE integer: the sum of the ranks' checksums must be the checksum of the
            # single-process solution of the whole job (the oracle on rank 0)
            chk = torch.from_numpy(np.array([avk_dist.result_checksum(batch, got)], np.uint64).view(np.int64).copy()).to(dev)
            dist.all_reduce(chk, op=dist.ReduceOp.SUM)
            if rank == 0:
                whole = oracle_lib.compare_batch(lib, job_batch, cs, threads=cpus, group_metrics=False)
                if int(chk.cpu().numpy().view(np.uint64)[0]) != avk_dist.result_checksum(job_batch, whole):
                    bad.append("job_checksum")
        parity = "bit-identical" if not bad else "MISMATCH:" + ",".join(bad)
        ok = torch.tensor([0 if bad else 1], device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            print("PARITY FAILURE on rank %d: %s" % (rank, parity), file=sys.stderr)
            sys.exit(3)
        log("parity vs oracle: %s (%d regions, oracle %.1f s on %d threads)" % (parity, n_regions, cpu_rate_parity[0], cpus))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_regions * args.steps / elapsed
        alg_bytes = ctx.algorithmic_bytes(batch)
        s_ms = float(np.mean(solver_ms))
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (s_ms * 1e-3) / 1e9
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] stand-in: HG002-scale SNV+indel compare, 24 contigs of GRCh38 primary lengths x %.3g (%d bases), "
                                   "%d regions in the job (%d on rank 0), %d calls on rank 0; seeds 20250103/4; inputs resident in HBM"
                                   % (args.scale, sum(c.size for c in contigs), n_job_regions if args.scaling == "strong" else total_regions, n_regions, batch.n_variants),
                       "regions_per_gpu": n_regions, "max_branch_factor": cfg.max_branch_factor, "min_variant_gap": 50,
                       "outputs_per_step": "per-region record (status, ed_h1, ed_h2, optima, types), per-variant decision word, 288-counter tally",
                       "parallelism": ("regions of ONE call set sharded by hash(region_id) over %d GPU(s)" % world if args.scaling == "strong" else
                                       "one call set per GPU on %d GPU(s)" % world) +
                                      ", no data-path collective; one RCCL all-reduce of the job tally (288 x int64) inside the timed region",
                       "parity": parity, "workspace_tiers": tiers, "lane_kernel_regions": lane_regions},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "all solver launches of a step (bulk + solo + overflow; HIP events ev0..ev1 on the launch stream)",
                         "kernel_ms": s_ms, "first_launch_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_region": alg_bytes / max(n_regions, 1)},
        }
        if boundary:
            out["host_boundary"] = boundary
        if byte_compares:
            out["dwfa_byte_compares_per_s"] = byte_compares * args.steps / elapsed * (total_regions / max(n_regions, 1))
            out["dwfa_byte_compares_per_region"] = byte_compares / max(n_regions, 1)
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib
            lib = oracle_lib.load()
            cs = oracle_lib.ContigSet(contigs)
            # all usable cores, >= 5 s of wall time: whole passes over the batch (same outputs as the GPU step: per-region records,
            # per-variant decisions, tally; every thread has solved regions before the clock of a pass matters: a pass takes seconds)
            est = cpu_rate_parity[1] / cpu_rate_parity[0] if cpu_rate_parity else 1e6
            reps = max(1, int(np.ceil(6.0 * est / max(n_regions, 1))))
            oracle_lib.compare_batch(lib, batch.slice(0, min(n_regions, 200_000)), cs, threads=cpus, group_metrics=False)  # warm-up
            tc = time.perf_counter()
            for _ in range(reps):
                oracle_lib.compare_batch(lib, batch, cs, threads=cpus, group_metrics=False)
            sec = time.perf_counter() - tc
            rate = n_regions * reps / sec
            n1 = min(n_regions, 300_000)
            sub = batch.slice(0, n1)
            oracle_lib.compare_batch(lib, sub.slice(0, min(n1, 20_000)), cs, threads=1, group_metrics=False)
            t1 = time.perf_counter()
            oracle_lib.compare_batch(lib, sub, cs, threads=1, group_metrics=False)
            rate1 = n1 / (time.perf_counter() - t1)
            out["cpu_baseline"] = {"value": rate, "unit": "regions/s", "cores": cpus, "kind": "port",
                                   "one_thread_value": rate1, "parallel_efficiency": rate / (rate1 * cpus),
                                   "host": "%d logical CPUs visible, %d usable (affinity / cgroup quota)" % (os.cpu_count() or 0, cpus),
                                   "sample": "%d pass(es) over the same %d-region batch on %d threads, %.2f s wall, same outputs as the GPU step; "
                                             "1 thread: first %d regions" % (reps, n_regions, cpus, sec, n1)}
            log("cpu baseline: %.0f regions/s on %d threads (%.2f s), 1 thread %.0f" % (rate, cpus, sec, rate1))
        if world == 1 and not args.no_secondary:
            contig2, batch2 = synth.config_chr20_snv()
            ctx.upload_reference([contig2])
            rb2 = ctx.upload(batch2)
            ctx.set_option("accumulate_tally", 0)
            for _ in range(10):
                ctx.compare_resident(rb2, cfg, None)
            ctx.synchronize()
            n2 = 200
            t2 = time.perf_counter()
            for _ in range(n2):  # every step synchronised: a step of this size is a launch chain over five streams, and queued back to
                ctx.compare_resident(rb2, cfg, None)  # back the cross-stream event waits of consecutive steps cost more than they hide
                ctx.synchronize()
            e2 = time.perf_counter() - t2
            t3 = time.perf_counter()
            for _ in range(n2):
                ctx.compare_resident(rb2, cfg, None)
            ctx.synchronize()
            e3 = time.perf_counter() - t3
            out["secondary"] = {"workload": "BASELINE configs[1]: synthetic chr20, 50000 SNV-only truth vs query calls, %d regions, resident" % batch2.n_regions,
                                "value": batch2.n_regions * n2 / e2, "unit": "regions/s", "ms_per_step": e2 / n2 * 1e3, "steps": n2,
                                "mode": "one host synchronisation per step", "queued_value": batch2.n_regions * n2 / e3, "queued_ms_per_step": e3 / n2 * 1e3}
            rb2.free()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
