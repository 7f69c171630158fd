#!/usr/bin/env python3
"""Benchmark of the compare hot path on MI355X.

Workload (default): the stand-in for BASELINE.json configs[2] "HG002 GIAB v4.2.1 truth vs DeepVariant query, GRCh38, SNV+indel, 1xMI355X" — the
configuration the metric is quoted on; the real GIAB files are not available offline, so the call sets are synthetic (SURVEY.md 8d config 3,
aardvark_amd/synth.py::config_genome): 24 contigs with GRCh38 primary lengths (3.1 Gbp), 3.9 M truth calls (82 % SNV, 9 % insertion, 9 % deletion, 3 %
of the sites within 30 bp of another, 2 % multi-allelic, 5 % of the indels in a repeat run with the query record shifted by whole units), query = truth
with 1 % dropped / 0.5 % zygosity flips / 0.5 % ALT changes / 1 % extra calls, seeds 20250103/4: about 3.57 M regions.

One step = ONE CALL of avk_compare_batch: the reference's rayon loop over solve_compare_region (src/main.rs:251-268, src/waffle_solver.rs:122) on the
boundary SURVEY.md 8d and BASELINE.md section 2 define — the region batch in HOST memory (pinned arrays from avk_host_alloc, which is where the Rust side
would build its FlatBatch) -> H2D of the caller's arrays as they are -> packing kernels -> solver kernels -> unpacking kernel -> D2H -> per-region
records, per-variant decisions and the 288-counter tally in the caller's HOST arrays.  `value` = regions/s of the timed steps on that boundary.
In the same JSON line:
  resident_value  regions/s of avk_compare_resident on the same batch already packed in HBM (no copies in the timed region): the kernels by themselves;
  roofline        algorithmic bytes of the batch / duration of all solver launches of a step (HIP events on the launch stream);
  cpu_baseline    the CPU restatement of the reference algorithm (oracle/) on the usable host cores, same outputs, >= 6 s of wall time;
  secondary       configs[1] (synthetic chr20 SNV), two robustness mixes (dense calls; --min-variant-gap 1000) and configs[4] (merge of three call sets,
                  majority strategy, avk_merge_batch host -> host) on one GPU.

N > 1 ranks (launched by torch.distributed.run, one rank per GPU): `--scaling strong` (the default for N > 1, the shape north_star describes): ONE call
set, rank r solves the regions with hash(region_id) % N == r (aardvark_amd/dist.py), the reference is replicated, no data-path collective; the job
tally is summed over the ranks by ONE RCCL all-reduce inside the timed region and the shards' per-variant decisions are checked against the
single-process solution through an order-independent checksum.  `--scaling weak`: every rank owns a whole-genome call set of its own.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # before anything loads the HIP runtime: the solver's streams need queues of their own, also beside a communicator's (aardvark_amd/__init__.py)
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

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
    ap.add_argument("--steps", type=int, default=100, help="timed avk_compare_batch calls (host arrays -> host arrays)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--resident-steps", type=int, default=300, help="timed avk_compare_resident steps of the resident leg (0 = skip)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrinks the contigs (and the call counts with them); 1.0 = the named workload")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto", help="auto = strong for N > 1 (one call set sharded over the ranks)")
    ap.add_argument("--records-only", action="store_true", help="the timed calls return per-region records, per-call decisions and the tally only — round 5's `value` — instead of "
                                                                "the reference's whole return value (those plus the per-region BASEPAIR groups)")
    ap.add_argument("--wide-results", action="store_true", help="the timed calls fill the wide result arrays instead of the packed form")
    ap.add_argument("--form", choices=("packed", "compact"), default="packed", help="flat form of the region batch the timed calls hand over: packed (avk_packed_batch, 94 MB per genome) or compact (avk_compact_batch, 227 MB)")
    ap.add_argument("--pageable", action="store_true", help="caller arrays in ordinary memory (the library stages them through a pinned bounce buffer) instead of avk_host_alloc memory")
    ap.add_argument("--watchdog-seconds", type=int, default=900, help="a run that takes longer dumps every thread's stack to stderr and exits (0 = off)")
    ap.add_argument("--torch", action="store_true",
                    help="one GPU: load PyTorch as the N > 1 ranks do (the process then runs on the HIP runtime the wheel bundles instead of the system's)")
    ap.add_argument("--no-supervisor", action="store_true",
                    help="run in this process (default for N > 1 ranks and under a profiler): otherwise the single-GPU run is a child of a thin supervisor "
                         "that has not touched the GPU, ends a run that exceeds --watchdog-seconds and starts it once more")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-merge", action="store_true", help="skip the configs[4] merge leg of `secondary`")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-`compare` wall-clock leg of `secondary` (files on disk, the command-line tool in fresh processes)")
    ap.add_argument("--merge-scale", type=float, default=1.0)
    ap.add_argument("--secondary-scale", type=float, default=0.25, help="genome scale of the two robustness mixes of `secondary`")
    ap.add_argument("--no-parity", action="store_true", help="skip the bit-identity gate against the oracle")
    ap.add_argument("--traffic-json", default=None,
                    help="PMC-derived HBM bytes per step collected with rocprofv3 --pmc in a builder-side run (default: the newest profiles/rNN_pmc_traffic.json); it is "
                         "reported only when the file names the build (avk_source_hash) of the library this process has loaded")
    return ap.parse_args()


def supervise(args):
    """A thin parent that has not touched the GPU: runs the benchmark as a child, forwards SIGTERM / SIGINT to the child's process group, ends a child
    that exceeds the watchdog and starts it once more (belt: 1,012 fresh-process first steps of the library ran without a stall in round 3,
    profiles/r03_first_steps_1012.log; the one stuck start of round 2 was never reproduced)."""
    env = dict(os.environ, AVK_BENCH_CHILD="1")
    state = {"child": None}

    def forward(signum, _frame):
        c = state["child"]
        if c is not None and c.poll() is None:
            try:
                os.killpg(c.pid, signal.SIGKILL)
            except Exception:
                c.kill()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, forward)
    signal.signal(signal.SIGINT, forward)

    def child_setup():
        os.setsid()
        try:  # the child dies with the supervisor even if the supervisor is killed outright
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)  # PR_SET_PDEATHSIG
        except Exception:
            pass

    rc = 1
    limit = args.watchdog_seconds + 60 if args.watchdog_seconds > 0 else None
    for attempt in range(2):
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, preexec_fn=child_setup)
        state["child"] = child
        try:
            rc = child.wait(timeout=limit)
            break
        except subprocess.TimeoutExpired:
            print("[bench supervisor] attempt %d exceeded %d s: ending it%s" % (attempt + 1, limit, ", starting once more" if attempt == 0 else ""), file=sys.stderr, flush=True)
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except Exception:
                child.kill()
            child.wait()
            rc = 124
    sys.exit(rc)


class HipRuntime:
    """The few runtime calls a one-GPU run needs besides the library's own, on the HIP runtime libaardvark_amd.so is linked with (no PyTorch in the process)."""

    def __init__(self, device):
        import ctypes as C
        import aardvark_amd
        aardvark_amd.load_library()
        path = next((l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l), "libamdhip64.so")
        self.C, self.lib = C, C.CDLL(path)
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.check(self.lib.hipSetDevice(device))

    def check(self, rc):
        if rc != 0:
            raise RuntimeError("HIP runtime call failed: %d" % rc)

    def synchronize(self):
        self.check(self.lib.hipDeviceSynchronize())

    def zeros_i64(self, n):
        return _DevI64(self, n)


class _DevI64:
    """n int64 words of device memory (the running tally of the resident leg): the three things bench.py does with the torch tensor it stands in for"""

    def __init__(self, hip, n):
        self.hip, self.n, self.ptr = hip, n, hip.C.c_void_p()
        hip.check(hip.lib.hipMalloc(hip.C.byref(self.ptr), 8 * n))
        self.zero_()

    def data_ptr(self):
        return self.ptr.value

    def zero_(self):
        self.hip.check(self.hip.lib.hipMemset(self.ptr, 0, 8 * self.n))

    def cpu(self):
        return self

    def numpy(self):
        import numpy as np
        out = np.zeros(self.n, np.int64)
        self.hip.check(self.hip.lib.hipMemcpy(out.ctypes.data, self.ptr, 8 * self.n, 2))  # hipMemcpyDeviceToHost
        return out


class HostCollectives:
    """torch.distributed with device tensors carried through host memory (a backend without device support: gloo); everything else is torch.distributed's"""

    def __init__(self, dist):
        self._dist = dist

    def __getattr__(self, name):
        return getattr(self._dist, name)

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self._dist.all_reduce(h, op=op if op is not None else self._dist.ReduceOp.SUM)
        t.copy_(h)

    def all_gather(self, out, t):
        hs = [o.cpu() for o in out]
        self._dist.all_gather(hs, t.cpu())
        for o, h in zip(out, hs):
            o.copy_(h)


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

    profiled = any(k in os.environ for k in ("ROCPROFILER_REGISTER_LIBRARY", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and not args.no_supervisor and not profiled and os.environ.get("AVK_BENCH_CHILD") != "1":
        supervise(args)

    if args.watchdog_seconds > 0:  # a stuck run must end with evidence instead of holding the machine
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)

    # stdout carries exactly ONE line, the JSON of rank 0: everything else that writes to file descriptor 1 — RCCL prints a version
    # banner there when a communicator comes up — is sent to stderr; the JSON goes to the original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import ctypes as C
    import numpy as np

    # One rank on one GPU runs WITHOUT PyTorch in the process (--torch forces it): the wheel brings a HIP runtime of its own (torch/lib/libamdhip64.so, ROCm 7.0)
    # that the whole process would then run on instead of the system's (7.2) — on it the copies of the asynchronous boundary do not run beside the kernels
    # (6.7-7.0 ms per genome against 4.3) and the queued resident step is 7 % slower (profiles/r05_runtime_ab.txt).  N > 1 ranks use torch.distributed (RCCL).
    use_dist = world > 1 or os.environ.get("AVK_BENCH_FORCE_DIST") == "1"  # the second form exercises the collective path on one GPU
    use_torch = use_dist or args.torch
    torch = dist = None
    if use_torch:
        import torch
        import torch.distributed as dist

    import aardvark_amd
    from aardvark_amd import CompareConfig, synth
    from aardvark_amd import dist as avk_dist
    from aardvark_amd._abi import ResultBatch

    def log(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_begin, msg), file=sys.stderr, flush=True)

    t_begin = time.perf_counter()
    dev = None
    if os.environ.get("AVK_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if use_torch:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL ("nccl") is the product's transport.  AVK_BENCH_BACKEND=gloo with AVK_BENCH_ONE_DEVICE=1 runs the SAME N-rank code — sharding, the timed region, the
        # collectives' call sites, the parity gates — with every rank on GPU 0 and the sums carried by gloo through host memory: what a one-GPU box can check of --gpus N
        # (RCCL refuses two ranks on one device); tests/test_bench_contract.py
        backend = os.environ.get("AVK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
            dist = HostCollectives(dist)
    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")

    # ---- workload
    cpus = usable_cpus()
    seed_shift = 1000 * rank if scaling == "weak" else 0
    contigs, batch = synth.config_genome(scale=args.scale, seed_ref=20250103 + seed_shift, seed_query=20250104 + seed_shift,
                                         threads=max(1, min(8, cpus // max(1, world))))
    n_job_regions = batch.n_regions
    job_batch = batch  # strong scaling: every rank holds the job's call set and solves its hash shard of it
    if scaling == "strong" and world > 1:
        batch = avk_dist.gather_calls(avk_dist.shard_batch(batch, rank, world))  # the rank's regions with their own calls only: a rank copies its shard, not the job
    n_regions = batch.n_regions
    log("workload: %d contigs, %d bases, %d regions on this rank, %d calls" % (len(contigs), sum(c.size for c in contigs), n_regions, batch.n_variants))

    ctx = aardvark_amd.Context(local_rank)
    hip = None
    if use_torch:
        # launches go on torch's current stream — a stream of the process's own, not the legacy null stream: with torch in the process the same queued steps take
        # 2.96 ms on the null stream and 2.64 on any other (2.56-2.59 without torch; tools/gpu_resident_host.py, profiles/r05_runtime_ab.txt)
        stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(stream)
        ctx.set_stream(stream.cuda_stream)
    else:
        hip = HipRuntime(local_rank)  # device memory for the running tally and the device-wide synchronisation, on the runtime the library is linked with
    ctx.set_option("emit_group_metrics", 0)  # per-region records + per-variant decisions + the tally; no 1144-byte block per region
    for kv in os.environ.get("AVK_OPTS", "").split(","):  # tuning experiments: context options by name
        if "=" in kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.upload_reference(contigs)
    cfg = CompareConfig(enable_sequences=False)
    ccfg = cfg.c_struct()
    log("reference resident")

    def fence():
        if not use_torch:
            hip.synchronize()  # hipDeviceSynchronize: every stream of the device, as torch.cuda.synchronize
            return
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- the timed steps: avk_compare_batch, host arrays -> host arrays
    # the caller's flat batch in the library's compact form (avk_compact_batch: 20 B per region + 17 B per call; the library widens it on the device),
    # the results in the arrays of avk_result_batch; both in pinned memory unless --pageable
    from aardvark_amd import CompactBatch, PackedBatch
    cmp_b = CompactBatch.from_region_batch(batch)
    # the form `value` is measured on: the packed one (avk_packed_batch: 10 B per region + 5 B per call + the allele bytes, offsets computed on the device) when
    # the batch fits its narrow fields, else the compact one
    form = args.form
    if form == "packed":
        try:
            hb = PackedBatch.from_compact(cmp_b)
        except ValueError:
            form, hb = "compact", cmp_b
    else:
        hb = cmp_b
    wide = batch
    if not args.pageable:
        hb, wide = (ctx.pinned_packed(hb) if form == "packed" else ctx.pinned_compact(hb)), ctx.pinned_batch(batch)
        cmp_b = ctx.pinned_compact(cmp_b) if form == "packed" else hb
    # the results in the packed form (avk_result_batch::region_packed / var_packed: 8 B per region + 1 B per call, everything the wide arrays say) unless --wide-results
    res_form = False if args.wide_results else "only"
    # `value` is measured on the reference's WHOLE return value: solve_compare_region gives a CompareBenchmark with the region's GroupTypeMetrics
    # (src/data_types/compare_benchmark.rs:9-33), and its BASEPAIR groups cannot be rebuilt from the per-call decisions (waffle_solver.rs:384-445) — so the timed calls
    # return them too, in the packed form (avk_result_batch::bp_packed / bp_spilled / bp_groups); with the per-call decisions they give the region's whole 13 x 22 block
    # (avk_group_metrics_from_compact; checked against the oracle below).  --records-only (and the forms that cannot carry them) measure round 5's lighter call.
    full_result = form == "packed" and not args.pageable and not args.wide_results and not args.records_only
    res = ResultBatch(hb, sequences=False, group_metrics=False, packed=res_form) if args.pageable else ctx.pinned_results(hb, packed=res_form, bp_groups="packed" if full_result else False)
    cb, ro = hb.c_struct(), res.c_struct()
    entry_point = ctx.lib.avk_compare_packed if form == "packed" else ctx.lib.avk_compare_compact
    entry_name = "avk_compare_packed" if form == "packed" else "avk_compare_compact"

    def step():
        ctx._check(entry_point(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))

    job_tally_host = np.zeros(aardvark_amd.TALLY_LEN, np.uint64)
    tally = torch.zeros(aardvark_amd.TALLY_LEN, dtype=torch.int64, device=dev) if use_torch else None
    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.all_reduce(tally, op=dist.ReduceOp.SUM)  # warm the communicator up as well
    fence()
    call_end = np.zeros(args.steps + 1)  # the calls are synchronous: a timestamp behind each gives the spread of the calls inside the timed region
    t0 = call_end[0] = time.perf_counter()
    for k in range(args.steps):
        step()
        job_tally_host += res.tally  # SummaryWriter::add_comparison_benchmark over the job's batches (writers/summary.rs:146-163)
        call_end[k + 1] = time.perf_counter()
    if use_torch:
        tally.copy_(torch.from_numpy(job_tally_host.astype(np.int64)))
    if use_dist:
        dist.all_reduce(tally, op=dist.ReduceOp.SUM)  # RCCL over xGMI: 288 x int64, the job's only collective
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
    call_ms = np.diff(call_end) * 1e3
    call_stats = {"min": round(float(call_ms.min()), 3), "median": round(float(np.median(call_ms)), 3), "p95": round(float(np.percentile(call_ms, 95)), 3),
                  "max": round(float(call_ms.max()), 3)} if args.steps else None
    log("timed region: %d %s calls in %.3f s (%.2f ms per call; single calls %s)" % (args.steps, entry_name, elapsed, elapsed / max(args.steps, 1) * 1e3, call_stats))
    job_tally = tally.cpu().numpy().copy() if use_torch else job_tally_host.astype(np.int64)
    res_ref_rp, res_ref_vp, res_ref_tally = (res.region_packed.copy(), res.var_packed.copy(), res.tally.copy()) if res_form else (None, None, None)
    got_boundary = res if args.wide_results else res.expanded(ctx.lib, batch)  # the outputs of the last timed call (the packed form: expanded on the host, outside the timed region)
    # the same call with the results in the wide arrays (18 B per region + 4 B per call over PCIe)
    wide_results_entry = None
    if not args.wide_results:
        r2 = ResultBatch(hb, sequences=False, group_metrics=False) if args.pageable else ctx.pinned_results(hb)
        ro2 = r2.c_struct()
        ctx._check(entry_point(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro2)))
        n_w = max(3, min(args.steps, 20))
        fence()
        tw0 = time.perf_counter()
        for _ in range(n_w):
            ctx._check(entry_point(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro2)))
        fence()
        w_elapsed = time.perf_counter() - tw0
        if world > 1:
            t = torch.tensor([w_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w_elapsed = float(t.item())
        wide_results_entry = {"value": total_regions * n_w / w_elapsed, "unit": "regions/s", "ms_per_step": w_elapsed / n_w * 1e3, "steps": n_w,
                              "host_bytes_out_per_step": int(sum(getattr(r2, f).nbytes for f in ("status", "ed_h1", "ed_h2", "n_optima", "type_present", "var_expected", "var_observed",
                                                                                                  "var_class", "var_zyg"))),
                              "same_outputs_as_value_leg": r2.diff(got_boundary) == [],
                              "what": "the value leg's call with the results in the wide arrays of avk_result_batch instead of the packed form"}
        del r2
    # the same job as batches IN FLIGHT inside the one context (avk_compare_packed_submit / avk_wait): genome k + 1 is copied and packed while genome k is solved, the
    # results of k cross while k + 1 is packed.  The synchronous call stays `value`; this is what a caller with several batches gets.
    pipelined_entry = None
    if form == "packed" and not args.pageable and world == 1:
        # (records and calls: a call that returns the BASEPAIR groups reads their spill count back before its last copy and is solved inside the submit)
        sets = [(hb, ctx.pinned_results(hb, packed=res_form) if full_result else res), (ctx.pinned_packed(hb), ctx.pinned_results(hb, packed=res_form))]
        n_p = max(4, min(args.steps, 20))
        tk = ctx.submit_packed(sets[0][0], config=cfg, res=sets[0][1])  # warm-up in the timed pattern: the second batch in flight takes a second set of device buffers
        for k in range(1, 5):
            nxt = ctx.submit_packed(sets[k & 1][0], config=cfg, res=sets[k & 1][1])
            tk.wait()
            tk = nxt
        tk.wait()
        fence()
        tp0 = time.perf_counter()
        tk = ctx.submit_packed(sets[0][0], config=cfg, res=sets[0][1])
        for k in range(1, n_p):
            nxt = ctx.submit_packed(sets[k & 1][0], config=cfg, res=sets[k & 1][1])
            tk.wait()
            tk = nxt
        tk.wait()
        fence()
        p_elapsed = time.perf_counter() - tp0
        same = all(np.array_equal(r.region_packed, res_ref_rp) and np.array_equal(r.var_packed, res_ref_vp) and np.array_equal(r.tally, res_ref_tally) for _, r in sets) if res_form else \
            all(r.diff(got_boundary) == [] for _, r in sets)
        pipelined_entry = {"value": total_regions * n_p / p_elapsed, "unit": "regions/s", "ms_per_step": p_elapsed / n_p * 1e3, "steps": n_p, "in_flight": 2,
                           "same_outputs_as_value_leg": bool(same),
                           "what": "the value leg's batch %d times back to back through avk_compare_packed_submit / avk_wait, two in flight in one context: copies in beside the solve of "
                                   "the batch before, copies out beside the packing of the batch behind; host arrays to host arrays" % n_p}
        log("pipelined boundary: %.3f ms per genome-sized batch (%s)" % (pipelined_entry["ms_per_step"], "outputs identical" if same else "OUTPUTS DIFFER"))
        if not same:
            print("PARITY FAILURE in the pipelined leg", file=sys.stderr)
            sys.exit(3)
        del sets
    # the same call with the OTHER set of outputs: `value` returns the full CompareBenchmark (records + per-call decisions + packed BASEPAIR groups: compare_benchmark.rs:9-33;
    # the "all 8 types" loop of waffle_solver.rs:384-445), this leg leaves the groups out (round 5's `value`) — or the other way round under --records-only
    bp_entry = records_entry = None
    if form == "packed" and not args.pageable and not args.wide_results:
        from aardvark_amd.api import group_metrics_from_compact
        other = ctx.pinned_results(hb, packed=res_form, bp_groups=False if full_result else "packed")
        cbp = hb.c_struct()
        rbo = other.c_struct()
        ctx._check(entry_point(ctx.handle, C.byref(cbp), C.byref(ccfg), C.byref(rbo)))
        n_b = max(3, min(args.steps, 20))
        fence()
        tb0 = time.perf_counter()
        for _ in range(n_b):
            ctx._check(entry_point(ctx.handle, C.byref(cbp), C.byref(ccfg), C.byref(rbo)))
        fence()
        b_elapsed = time.perf_counter() - tb0
        if world > 1:
            t = torch.tensor([b_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            b_elapsed = float(t.item())
        b_solver_ms = ctx.last_solver_ms()
        rbp = res if full_result else other
        bp_ms, rec_ms = (elapsed / max(args.steps, 1) * 1e3, b_elapsed / n_b * 1e3) if full_result else (b_elapsed / n_b * 1e3, elapsed / max(args.steps, 1) * 1e3)
        spilled = int(rbp.bp_spilled[0])
        same = (np.array_equal(other.region_packed, res_ref_rp) and np.array_equal(other.var_packed, res_ref_vp)) if res_form else other.diff(got_boundary) == []
        alg_g = ctx.algorithmic_bytes(batch, with_groups=True)
        bp_entry = {"value": total_regions / (bp_ms * 1e-3), "unit": "regions/s", "ms_per_step": bp_ms, "steps": args.steps if full_result else n_b,
                    "is_the_value_leg": bool(full_result), "extra_ms_over_records_only": bp_ms - rec_ms,
                    "host_bytes_out_per_step_extra": 4 * n_regions + 16 * spilled, "bytes_per_region_extra": (4 * n_regions + 16 * spilled) / max(n_regions, 1),
                    "regions_in_one_word": int(n_regions - int(((rbp.bp_packed[:n_regions] & np.uint32(0x80000000)) != 0).sum())), "spilled_groups": spilled,
                    "same_records_and_calls_in_both_legs": bool(same),
                    "what": "the call returning per-region BASEPAIR groups as well (avk_result_batch::bp_packed / bp_spilled / bp_groups): with the per-call decisions they are the "
                            "region's whole GroupTypeMetrics (avk_group_metrics_from_compact), i.e. the reference's full return value"}
        if not full_result:
            bp_entry["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_launch": int(alg_g), "bytes_per_region": alg_g / max(n_regions, 1), "kernel_ms": b_solver_ms,
                                    "achieved": alg_g / max(b_solver_ms, 1e-9) / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": alg_g / max(b_solver_ms, 1e-9) / 1e6 / 8000.0}
        records_entry = {"value": total_regions / (rec_ms * 1e-3), "unit": "regions/s", "ms_per_step": rec_ms, "steps": n_b if full_result else args.steps,
                         "is_the_value_leg": not full_result,
                         "what": "the call returning per-region records (8 B), per-call decisions (1 B) and the tally only — round 5's `value`; the BASEPAIR groups of a region are not "
                                 "derivable from these"}
        if not same:
            print("PARITY FAILURE: the two result sets disagree on records / calls", file=sys.stderr)
            sys.exit(3)
        bp_check = (rbp, group_metrics_from_compact)
        log("full result (with BASEPAIR groups): %.3f ms per call; records and calls only: %.3f ms (%+.3f); %.2f extra bytes per region, %d groups spilled" % (
            bp_ms, rec_ms, bp_ms - rec_ms, bp_entry["bytes_per_region_extra"], spilled))
    # the same boundary with the batch in the compact form (avk_compact_batch: 20 B per region + 17 B per call, explicit offsets), when `value` is on the packed one
    compact_entry = None
    if form == "packed":
        cres = ResultBatch(cmp_b, sequences=False, group_metrics=False) if args.pageable else ctx.pinned_results(cmp_b)
        ccb, cro = cmp_b.c_struct(), cres.c_struct()
        ctx._check(ctx.lib.avk_compare_compact(ctx.handle, C.byref(ccb), C.byref(ccfg), C.byref(cro)))
        n_c = max(3, min(args.steps, 20))
        fence()
        tc0 = time.perf_counter()
        for _ in range(n_c):
            ctx._check(ctx.lib.avk_compare_compact(ctx.handle, C.byref(ccb), C.byref(ccfg), C.byref(cro)))
        fence()
        c_elapsed = time.perf_counter() - tc0
        if world > 1:
            t = torch.tensor([c_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c_elapsed = float(t.item())
        same = cres.diff(res) == []
        compact_entry = {"value": total_regions * n_c / c_elapsed, "unit": "regions/s", "ms_per_step": c_elapsed / n_c * 1e3, "steps": n_c, "host_bytes_in_per_step": cmp_b.nbytes(),
                         "same_outputs_as_value_leg": same, "what": "the same boundary with the batch in the compact form (avk_compact_batch through avk_compare_compact)"}
        del cres
    # the same boundary with the batch in the WIDE structure-of-arrays form (avk_region_batch: 52 B per region + 38 B per call over PCIe)
    wres = ResultBatch(wide, sequences=False, group_metrics=False) if args.pageable else ctx.pinned_results(wide)
    wcb, wro = wide.c_struct(), wres.c_struct()
    ctx._check(ctx.lib.avk_compare_batch(ctx.handle, C.byref(wcb), C.byref(ccfg), C.byref(wro)))
    n_wide = max(3, min(args.steps, 20))
    fence()
    tw = time.perf_counter()
    for _ in range(n_wide):
        ctx._check(ctx.lib.avk_compare_batch(ctx.handle, C.byref(wcb), C.byref(ccfg), C.byref(wro)))
    fence()
    wide_elapsed = time.perf_counter() - tw
    if world > 1:
        t = torch.tensor([wide_elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wide_elapsed = float(t.item())

    # ---- resident leg: the same batch packed once in HBM, avk_compare_resident steps queued back to back (the kernels by themselves)
    resident = None
    rb = ctx.upload(batch)
    kernel_ms, solver_ms = [], []
    if full_result:
        ctx.set_option("emit_bp_groups", 1)  # the resident steps (and the roofline's kernel durations) write what the value leg's steps write: the per-region groups too
    if args.resident_steps > 0:
        ctx.set_option("accumulate_tally", 1)
        rtally = torch.zeros(aardvark_amd.TALLY_LEN, dtype=torch.int64, device=dev) if use_torch else hip.zeros_i64(aardvark_amd.TALLY_LEN)
        for _ in range(args.warmup):
            ctx.compare_resident(rb, cfg, rtally.data_ptr())
        fence()
        rtally.zero_()
        fence()
        tr = time.perf_counter()
        for _ in range(args.resident_steps):
            ctx.compare_resident(rb, cfg, rtally.data_ptr())
        if use_dist:
            dist.all_reduce(rtally, op=dist.ReduceOp.SUM)
        fence()
        r_elapsed = time.perf_counter() - tr
        if world > 1:
            t = torch.tensor([r_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r_elapsed = float(t.item())
        resident = {"value": total_regions * args.resident_steps / r_elapsed, "unit": "regions/s", "ms_per_step": r_elapsed / args.resident_steps * 1e3,
                    "steps": args.resident_steps, "what": "avk_compare_resident on the batch packed once in HBM, steps queued back to back, tally added on the device; no copies in the timed region"}
        resident_tally = rtally.cpu().numpy().copy()
        ctx.set_option("accumulate_tally", 0)
        log("resident leg: %.3f ms per step" % resident["ms_per_step"])
    # kernel durations for the roofline: HIP events the library records on the launch stream, read on steps outside the timed regions
    for _ in range(5):
        ctx.compare_resident(rb, cfg, None)
        kernel_ms.append(ctx.last_kernel_ms())
        solver_ms.append(ctx.last_solver_ms())
    ctx.set_option("emit_bp_groups", 0)
    tiers = lane_regions = wide_regions = None
    # the algorithmic bytes of every LAUNCH CLASS (SURVEY 8d's per-region formula over the regions the class's launches take: the library's own work order), so that a
    # per-kernel fraction can be recomputed from the rocprof durations of profiles/
    by_class = None
    try:
        order, plan = ctx.work_order(rb)
        n_c, n_b, n_l = plan["class_c"], plan["class_b"], plan["lanes"]
        names = ["one call per side, <= 112 bases", "one call per side, <= 192 bases", "two calls per side, <= 160 bases", "two calls per side, <= 192 bases",
                 "three calls per side", "looked-up pairs (the same SNV on both sides)"]
        spans = [("class C (large searches: wide kernel, long windows)", 0, n_c), ("class B (solo launches)", n_c, n_b), ("bulk (wave-per-region, LDS tiers)", n_c + n_b, n_regions - n_l - n_c - n_b)]
        for k, (first, cnt, heavy) in enumerate(plan["fast"]):
            if cnt and k == 4:  # the whole three-call class runs four lanes per region (16 records per wave)
                spans.append(("lanes: %s (quads)" % names[k], first, cnt))
            elif cnt and k < 5:
                spans.append(("lanes: %s — head (quads)" % names[k], first, heavy))
                spans.append(("lanes: %s — rest (64 per wave)" % names[k], first + heavy, cnt - heavy))
            elif cnt:
                spans.append(("lanes: %s" % names[k], first, cnt))
        by_class = []
        for name, first, cnt in spans:
            if cnt <= 0:
                continue
            sub = avk_dist.take_regions(batch, np.sort(order[first:first + cnt].astype(np.int64)))
            by_class.append({"launch_class": name, "regions": int(cnt), "algorithmic_bytes": int(ctx.algorithmic_bytes(sub, with_groups=bool(full_result)))})
    except Exception as e:  # a measurement aid: its absence must not cost the run
        by_class = "unavailable: %s" % e
    got = ctx.download(rb, group_metrics=False)
    try:
        tiers = ctx.last_tier_counts()  # regions finished per workspace tier of the wave-per-region kernels, then capacity failures
        lane_regions = ctx.last_lane_solved()  # regions finished by the lane-per-region kernel
        wide_regions = ctx.last_wide_solved()  # regions finished by the wave-cooperative kernel of the large searches on small windows
    except Exception:
        pass
    rb.free()

    # ---- bit-identity gate (same-run rule): this rank's outputs — of the host boundary AND of the resident path — against the oracle
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
        bad = ["boundary:" + x for x in got_boundary.diff(want)] + ["wide_boundary:" + x for x in wres.diff(want)] + ["resident:" + x for x in got.diff(want)]
        if bp_entry is not None:  # the full 13 x 22 block of every region of the first contig, rebuilt from the packed groups + the per-call decisions, against the oracle's
            n0 = int((batch.contig_idx == batch.contig_idx[0]).sum()) if batch.contig_idx is not None else min(n_regions, 200_000)
            n0 = min(n0, 300_000)
            sub = batch.slice(0, n0)  # (shares the call arrays: the results' per-call arrays are indexed as the whole batch's)
            if True:
                want0 = oracle_lib.compare_batch(lib, sub, cs, threads=cpus, group_metrics=True)
                rbp, from_compact = bp_check
                exp = rbp.expanded(ctx.lib, batch) if res_form else rbp
                exp.bp_packed, exp.bp_spilled, exp.bp_groups, exp.bp_off = rbp.bp_packed, rbp.bp_spilled, rbp.bp_groups, None
                full = from_compact(sub, exp)
                ok0 = want0.status == 0
                if not np.array_equal(full[ok0], want0.group_metrics[ok0]):
                    bad.append("bp_groups_full_blocks")
                bp_entry["parity"] = "the 13 x 22 blocks of the first %d regions (one contig) rebuilt from the packed groups equal the oracle's" % n0 if "bp_groups_full_blocks" not in bad else "MISMATCH"
        # the job total must be steps x this rank's tally, summed over the ranks
        mine = want.tally.astype(np.int64)
        if use_dist:
            mine_d = torch.from_numpy(mine).to(dev)
            dist.all_reduce(mine_d, op=dist.ReduceOp.SUM)
            mine = mine_d.cpu().numpy()
        if not np.array_equal(mine * args.steps, job_tally):
            bad.append("job_tally")
        if resident is not None and not np.array_equal(mine * args.resident_steps, resident_tally):
            bad.append("resident_job_tally")
        if scaling == "strong" and world > 1:
            # the shards' per-variant decisions, gathered as ONE integer: the sum of the ranks' checksums must be the checksum of the
            # single-process solution of the whole job (the oracle on rank 0)
            chk = torch.from_numpy(np.array([avk_dist.result_checksum(batch, got_boundary)], np.uint64).view(np.int64).copy()).to(dev)
            dist.all_reduce(chk, op=dist.ReduceOp.SUM)
            if rank == 0:
                whole = oracle_lib.compare_batch(lib, job_batch, cs, threads=cpus, group_metrics=False)
                if int(chk.cpu().numpy().view(np.uint64)[0]) != avk_dist.result_checksum(job_batch, whole):
                    bad.append("job_checksum")
        parity = "bit-identical" if not bad else "MISMATCH:" + ",".join(bad)
        ok_all = 0 if bad else 1
        if world > 1:
            ok = torch.tensor([ok_all], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            ok_all = int(ok.item())
        if ok_all != 1:
            print("PARITY FAILURE on rank %d: %s" % (rank, parity), file=sys.stderr)
            sys.exit(3)
        log("parity vs oracle: %s (%d regions, oracle %.1f s on %d threads)" % (parity, n_regions, cpu_rate_parity[0], cpus))

    # ---- BASELINE configs[4] on N > 1 GPUs: the merge job cut by the same rule (every rank takes part; rank 0 reports)
    merge_sharded = None
    if world > 1 and not args.no_merge and not args.no_secondary:
        merge_sharded = merge_leg_sharded(ctx, args, rank, world, torch, dist, dev, cpus, log, fence)
        ctx.upload_reference(contigs)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_regions * args.steps / elapsed
        alg_bytes = ctx.algorithmic_bytes(batch, with_groups=bool(full_result))  # per-region records, per-call decisions, the tally — and the per-region groups when the steps write them
        s_ms = float(np.mean(solver_ms))
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (s_ms * 1e-3) / 1e9
        # counter traffic is a builder-side measurement (rocprofv3 --pmc needs its own runs): it is reported only for the build it was measured on
        traffic = None
        tj = args.traffic_json
        if tj is None:
            found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
            tj = found[-1] if found else None
        loaded_hash = (ctx.lib.avk_source_hash() or b"").decode()
        if tj is None or not os.path.exists(tj):
            traffic_src = "null: no profiles/rNN_pmc_traffic.json"
        else:
            try:
                tjs = json.load(open(tj))
                if tjs.get("source_hash") != loaded_hash or loaded_hash in ("", "unknown"):
                    traffic_src = "null: %s was measured on build %s, the loaded library is build %s" % (os.path.relpath(tj, ROOT), tjs.get("source_hash", "(unnamed)"), loaded_hash)
                else:
                    traffic = tjs.get("hbm_bytes_per_launch")
                    traffic_src = ("NOT measured in this process: FETCH_SIZE + WRITE_SIZE of a builder-side `rocprofv3 --pmc` run of the resident step on this same build (%s), %s"
                                   % (loaded_hash, os.path.relpath(tj, ROOT)))
            except Exception as e:
                traffic_src = "null: %s unreadable (%s)" % (os.path.relpath(tj, ROOT), e)
        in_bytes = hb.nbytes()
        wide_bytes = sum(getattr(wide, f).nbytes for f in ("contig_idx", "start", "end", "t_off", "t_cnt", "q_off", "q_cnt", "var_pos", "var_type", "var_zyg", "var_raw_space",
                                                            "a0_off", "a0_len", "a1_off", "a1_len", "allele_bytes"))
        out_bytes = int(sum(getattr(res, f).nbytes for f in ("status", "ed_h1", "ed_h2", "n_optima", "type_present", "var_expected", "var_observed", "var_class", "var_zyg",
                                                             "region_packed", "var_packed") if getattr(res, f) is not None))
        out = {
            "metric": "compared regions/sec (whole node)",
            "value": value,
            "unit": "regions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] stand-in: HG002-scale SNV+indel compare, 24 contigs of GRCh38 primary lengths x %.3g (%d bases), "
                                   "%d regions in the job (%d on rank 0), %d calls on rank 0; seeds 20250103/4; one step = one %s call, region batch (%s flat form) and "
                                   "results in %s host memory (the reference loop's boundary, src/main.rs:251-268), H2D and D2H inside the timed region"
                                   % (args.scale, sum(c.size for c in contigs), n_job_regions if scaling == "strong" else total_regions, n_regions, batch.n_variants, entry_name, form,
                                      "pageable" if args.pageable else "pinned (avk_host_alloc)"),
                       "regions_per_gpu": n_regions, "max_branch_factor": cfg.max_branch_factor, "min_variant_gap": 50,
                       "host_bytes_in_per_step": in_bytes, "host_bytes_out_per_step": out_bytes,
                       "outputs_per_step": ("the reference's full return value (CompareBenchmark, compare_benchmark.rs:9-33): " if full_result else "") +
                                           "per-region record (status, ed_h1, ed_h2, optima, types), per-variant decision (EA, OA, class, resolved zygosity), 288-counter tally" +
                                           ("; per-region BASEPAIR groups, packed: one word per region, the groups of multi-type regions spilled (with the per-call decisions: the "
                                            "region's whole 13 x 22 GroupTypeMetrics block, avk_group_metrics_from_compact)" if full_result else "; NO per-region groups (--records-only)") +
                                           ("" if args.wide_results else "; regions and calls in the packed form of avk_result_batch (8 B + 1 B; avk_results_expand gives the wide arrays)"),
                       "parallelism": ("regions of ONE call set sharded by hash(region_id) over %d GPU(s)" % world if scaling == "strong" else
                                       "one call set per GPU on %d GPU(s)" % world) +
                                      ", no data-path collective; one RCCL all-reduce of the job tally (288 x int64) inside the timed region",
                       "parity": parity, "workspace_tiers": tiers, "lane_kernel_regions": lane_regions, "wide_kernel_regions": wide_regions},
            "call_ms": call_stats,  # this rank's single calls inside the timed region (value = all of them, barrier to barrier)
            "resident_value": resident["value"] if resident else None,
            "resident": resident,
            "wide_results": wide_results_entry,
            "pipelined": pipelined_entry,
            "with_bp_groups": bp_entry,
            "records_only": records_entry,
            "process": ("PyTorch loaded: the process runs on the HIP runtime the wheel bundles (torch/lib/libamdhip64.so)" if use_torch else
                        "one rank, PyTorch not loaded: the process runs on the system's HIP runtime, the one libaardvark_amd.so is linked with (--torch loads it as N > 1 ranks do)"),
            "compact_soa": compact_entry,
            "wide_soa": {"value": total_regions * n_wide / wide_elapsed, "unit": "regions/s", "ms_per_step": wide_elapsed / n_wide * 1e3, "steps": n_wide, "host_bytes_in_per_step": wide_bytes,
                         "what": "the same boundary with the batch in the wide structure-of-arrays form (avk_region_batch through avk_compare_batch)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "all solver launches of a step (lane classes + looked-up pairs + wide + bulk + solo + overflow; HIP events ev0..ev1 on the launch stream)",
                         "kernel_ms": s_ms, "first_launch_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_by_launch_class": by_class,
                         "bytes_per_region": alg_bytes / max(n_regions, 1),
                         "pcie": {"bytes_per_step": in_bytes + out_bytes, "achieved_GBs": (in_bytes + out_bytes) * (total_regions / max(n_regions, 1)) / world / (ms_per_step * 1e-3) / 1e9,
                                  "note": "the host boundary moves the caller's arrays over PCIe (57 GB/s each way measured on this pool, profiles/r03_pcie_probe.txt): its floor per step is bytes / 57 GB/s"}},
        }
        if byte_compares:  # base comparisons the reference algorithm makes on this batch (counted by the oracle), at the rates above
            out["dwfa_byte_compares_per_region"] = byte_compares / max(n_regions, 1)
            out["dwfa_byte_compares_per_s"] = out["dwfa_byte_compares_per_region"] * value
            if resident:
                out["dwfa_byte_compares_per_s_resident"] = out["dwfa_byte_compares_per_region"] * resident["value"]
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib
            lib = oracle_lib.load()
            cs = oracle_lib.ContigSet(contigs)
            # all usable cores, >= 6 s of wall time: whole passes over the batch (same outputs as the GPU step: per-region records,
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
                                   "all_core_extrapolation": {"value": rate / cpus * (os.cpu_count() or cpus), "logical_cpus": os.cpu_count() or cpus,
                                                              "note": "NOT measured: the rate above scaled from the %d granted cores to the %d logical CPUs the box shows, at the measured per-core rate "
                                                                      "(an upper bound: hyper-threads and memory bandwidth scale worse); north_star's '10x the all-core CPU rate' is checked against this"
                                                                      % (cpus, os.cpu_count() or cpus),
                                                              "gpu_over_this": value / max(rate / cpus * (os.cpu_count() or cpus), 1.0)},
                                   "host": "%d logical CPUs visible, %d usable (affinity / cgroup quota)" % (os.cpu_count() or 0, cpus),
                                   "sample": "%d pass(es) over the same %d-region batch on %d threads, %.2f s wall, same outputs as the GPU step, host arrays in and out; "
                                             "1 thread: first %d regions" % (reps, n_regions, cpus, sec, n1)}
            log("cpu baseline: %.0f regions/s on %d threads (%.2f s), 1 thread %.0f" % (rate, cpus, sec, rate1))
        if merge_sharded is not None:
            out["secondary"] = {"merge_3_callers": merge_sharded}
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_legs(ctx, cfg, args, cpus, log, contigs, job_batch, ms_per_step, resident["ms_per_step"] if resident else None)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def secondary_legs(ctx, cfg, args, cpus, log, job_contigs=None, job_batch=None, headline_ms=None, headline_resident_ms=None):
    """configs[1], the two robustness mixes and configs[4] on one GPU: each with the host-boundary rate (pinned arrays), the resident step and — where it
    says so — a parity check against the oracle"""
    import ctypes as C
    import numpy as np
    import oracle_lib
    from aardvark_amd import synth
    lib = oracle_lib.load()
    ccfg = cfg.c_struct()
    sec = {}

    def cpu_figure(fn, n_units, unit, min_seconds=2.0):
        """the oracle on the usable cores beside a leg: whole passes until min_seconds have gone by (a reported baseline, not a target)"""
        t0 = time.perf_counter()
        passes = 0
        while True:
            out = fn()
            passes += 1
            if time.perf_counter() - t0 >= min_seconds:
                break
        dt = time.perf_counter() - t0
        return out, {"value": n_units * passes / dt, "unit": unit, "cores": cpus, "kind": "port", "sample": "%d pass(es) over the leg's batch on %d threads, %.2f s" % (passes, cpus, dt)}

    def compare_leg(name, contigs, batch, what, parity=True, resident_sync=False, few=False, packed=False):
        from aardvark_amd import CompactBatch, PackedBatch
        ctx.upload_reference(contigs)
        if packed:  # the form the headline value is measured on
            hb = ctx.pinned_packed(PackedBatch.from_compact(CompactBatch.from_region_batch(batch)))
            entry_point = ctx.lib.avk_compare_packed
        else:
            hb = ctx.pinned_batch(batch)
            entry_point = ctx.lib.avk_compare_batch
        res = ctx.pinned_results(hb, packed="only" if packed else False)  # (the packed batch form goes with the packed result form, as on the value leg)
        cb, ro = hb.c_struct(), res.c_struct()
        call = lambda: ctx._check(entry_point(ctx.handle, C.byref(cb), C.byref(ccfg), C.byref(ro)))
        call()
        t1 = time.perf_counter()
        call()
        one = time.perf_counter() - t1
        nb = 2 if few else max(3, min(200, int(0.5 / max(one, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(nb):
            call()
        be = time.perf_counter() - t0
        rb = ctx.upload(batch)
        for _ in range(1 if few else 3):
            ctx.compare_resident(rb, cfg, None)
        ctx.synchronize()
        nr = 2 if few else max(5, min(400, int(0.5 / max(one, 1e-4)) * 2))
        t0 = time.perf_counter()
        for _ in range(nr):
            ctx.compare_resident(rb, cfg, None)
        ctx.synchronize()
        re_ = time.perf_counter() - t0
        got = ctx.download(rb, group_metrics=False)
        entry = {"workload": what, "regions": batch.n_regions, "value": batch.n_regions * nb / be, "unit": "regions/s", "ms_per_step": be / nb * 1e3, "steps": nb,
                 "resident_value": batch.n_regions * nr / re_, "resident_ms_per_step": re_ / nr * 1e3, "lane_kernel_regions": ctx.last_lane_solved(),
                 "lane_share": ctx.last_lane_solved() / max(batch.n_regions, 1), "wide_kernel_regions": ctx.last_wide_solved(), "workspace_tiers": ctx.last_tier_counts(),
                 "boundary_form": "packed batch and packed results (avk_compare_packed)" if packed else "wide (avk_compare_batch)"}
        if resident_sync:  # every step synchronised: a step of this size is a launch chain over several streams
            t0 = time.perf_counter()
            for _ in range(nr):
                ctx.compare_resident(rb, cfg, None)
                ctx.synchronize()
            entry["resident_sync_ms_per_step"] = (time.perf_counter() - t0) / nr * 1e3
        rb.free()
        if parity:
            cs_leg = oracle_lib.ContigSet(contigs)
            want, entry["cpu_baseline"] = cpu_figure(lambda: oracle_lib.compare_batch(lib, batch, cs_leg, threads=cpus, group_metrics=False), batch.n_regions, "regions/s")
            entry["gpu_over_cpu"] = {"boundary": entry["value"] / entry["cpu_baseline"]["value"], "resident": entry["resident_value"] / entry["cpu_baseline"]["value"]}
            bad = ["boundary:" + x for x in (res.expanded(ctx.lib, batch) if packed else res).diff(want)] + ["resident:" + x for x in got.diff(want)]
            entry["parity"] = "bit-identical" if not bad else "MISMATCH:" + ",".join(bad)
            if bad:
                print("PARITY FAILURE in secondary leg %s: %s" % (name, entry["parity"]), file=sys.stderr)
                sys.exit(3)
        sec[name] = entry
        log("secondary %s: boundary %.3f ms, resident %.3f ms per step, lane share %.4f" % (name, entry["ms_per_step"], entry["resident_ms_per_step"], entry["lane_share"]))

    if job_batch is not None:
        # What ONE rank of an 8-GPU strong-scaling run does per step: rank 0's hash shard of the job (aardvark_amd/dist.py::shard_batch, gather_calls), same call,
        # same outputs.  NOT a scaling curve (this pool has one GPU; the driver measures the curve when it has a node): the per-rank step that bounds N = 8.
        from aardvark_amd import dist as avk_dist
        shard = avk_dist.gather_calls(avk_dist.shard_batch(job_batch, 0, 8))
        compare_leg("shard_1_of_8", job_contigs, shard, "rank 0's hash shard (region_id %% 8 == 0 after hashing) of the headline job: %d of %d regions" % (shard.n_regions, job_batch.n_regions), packed=True)
        sec["shard_1_of_8"]["strong_scaling_bound"] = {
            "boundary": headline_ms / sec["shard_1_of_8"]["ms_per_step"] if headline_ms else None,
            "resident": headline_resident_ms / sec["shard_1_of_8"]["resident_ms_per_step"] if headline_resident_ms else None,
            "note": "the most an 8-GPU strong-scaling run can gain over one GPU = the headline ms_per_step / this leg's (the ranks run side by side, one all-reduce of 288 counters "
                    "behind them); measured on ONE GPU, no scaling curve can be measured on this pool"}
    contig2, batch2 = synth.config_chr20_snv()
    compare_leg("chr20_snv", [contig2], batch2, "BASELINE configs[1]: synthetic chr20, 50000 SNV-only truth vs query calls, %d regions" % batch2.n_regions, resident_sync=True)
    # robustness: a denser, messier mix and the reference's recommended SV / TR setting (docs/recommended_settings.md:16-37)
    contigs3, batch3 = synth.config_genome(scale=args.secondary_scale, threads=min(8, cpus), close_frac=0.10, str_frac=0.15, multi_frac=0.05)
    compare_leg("dense_mix", contigs3, batch3, "genome x %.3g, 10 %% of the sites within 30 bp of another, 15 %% of the indels in repeat runs, 5 %% multi-allelic, %d regions" % (args.secondary_scale, batch3.n_regions))
    # a harder, more honest reference: the headline genome is i.i.d. uniform ACGT — the friendliest sequence an aligner can get.  This one has GRCh38's share of
    # simple repeats (homopolymer, di- and tri-nucleotide tracts, one per 700 bases), assembly gaps of N, and half of its indels inside repeat runs (written at shifted
    # positions on the query side).  Its soft mask is folded to upper case, as the tool folds it when it loads a FASTA (--reference-case upper).
    contigs6, batch6 = synth.config_genome(scale=args.secondary_scale, threads=min(8, cpus), low_complexity=True, str_frac=0.5)
    compare_leg("lowcomplexity_genome", contigs6, batch6, "genome x %.3g with a low-complexity background (4 %% of the bases in simple-repeat tracts, N gaps outside the confident intervals), "
                "half of the indels in repeat runs, %d regions" % (args.secondary_scale, batch6.n_regions), packed=True)
    n6 = max(batch6.n_regions, 1)
    sec["lowcomplexity_genome"]["looked_up_share"] = None  # (the lanes' count includes the looked-up class; the hand-back share is what the wide and wave kernels finished)
    sec["lowcomplexity_genome"]["handed_back_or_outside_lanes_share"] = 1.0 - sec["lowcomplexity_genome"]["lane_share"]
    sec["lowcomplexity_genome"]["n_bases_share"] = float(sum(int((c == ord("N")).sum()) for c in contigs6) / max(sum(c.size for c in contigs6), 1))
    del contigs6, batch6
    # real call sets, when the host has them: AVK_REAL_REF / AVK_REAL_TRUTH / AVK_REAL_QUERY / AVK_REAL_BED (FASTA[.gz], two VCF.gz, confident BED — e.g. GRCh38, HG002 GIAB
    # v4.2.1 and a DeepVariant call set, north_star's target).  This pool has no network: none can be fetched here.
    real = {k: os.environ.get("AVK_REAL_" + k) for k in ("REF", "TRUTH", "QUERY", "BED")}
    if all(real.values()):
        from aardvark_amd import feeder
        t_feed = time.perf_counter()
        genome = feeder.Genome(real["REF"])
        feed = feeder.feed_compare(real["TRUTH"], real["QUERY"], real["BED"], genome)
        feed_s = time.perf_counter() - t_feed
        compare_leg("real_data", genome.contigs(), feed.batch, "files named by AVK_REAL_REF/TRUTH/QUERY/BED through the feeder library: %d regions (%.1f s to load and walk)" % (feed.batch.n_regions, feed_s))
        sec["real_data"]["files"] = real
    else:
        sec["real_data"] = {"skipped": "not supplied: set AVK_REAL_REF, AVK_REAL_TRUTH, AVK_REAL_QUERY and AVK_REAL_BED to run real call sets (GRCh38 + HG002 GIAB v4.2.1 vs a DeepVariant "
                                       "query is north_star's target; no network on this pool)"}
    # (windows of kilobases: every region goes to the wave-per-region kernels and costs some fifty times a small one — on the CPU as well; a fifth of the other leg's scale)
    contigs4, batch4 = synth.config_genome(scale=args.secondary_scale / 5, threads=min(8, cpus), gap=1000)
    compare_leg("min_variant_gap_1000", contigs4, batch4, "genome x %.3g clustered with --min-variant-gap 1000, %d regions" % (args.secondary_scale / 5, batch4.n_regions), few=True)
    if not args.no_merge:
        from aardvark_amd.merge import MergeConfig, MultiBatch, PackedMultiBatch, merge_multi_batch, pinned_multi_batch
        import merge_oracle as mo
        contigs5, mb = synth.config_genome_merge(scale=args.merge_scale, k=3, threads=min(8, cpus))
        ctx.upload_reference(contigs5)
        wide_mb = mb
        try:  # the packed form (avk_merge_packed) when the batch fits it, as the compare legs do
            timed_mb = PackedMultiBatch.from_multi(mb) if args.form == "packed" else mb
        except ValueError:
            timed_mb = mb
        merge_packed = isinstance(timed_mb, PackedMultiBatch)
        if not args.pageable:
            timed_mb, wide_mb = pinned_multi_batch(ctx, timed_mb), (pinned_multi_batch(ctx, mb) if merge_packed else None)
            if wide_mb is None:
                wide_mb = timed_mb
        mcfg = MergeConfig(majority_voting_enabled=True)
        merge_multi_batch(ctx, timed_mb, mcfg)
        t0 = time.perf_counter()
        nm = 6
        for _ in range(nm):
            mres = merge_multi_batch(ctx, timed_mb, mcfg)
        me = time.perf_counter() - t0
        where = "pageable" if args.pageable else "pinned"
        entry = {"workload": "BASELINE configs[4] stand-in on ONE GPU: merge of 3 call sets (seeds 20250105-7) x %.3g genome, majority strategy, %d regions, %d input pairs"
                             % (args.merge_scale, mb.n_regions, 3 * mb.n_regions),
                 "value": mb.n_regions * nm / me, "unit": "merge regions/s", "ms_per_step": me / nm * 1e3, "steps": nm,
                 "bytes_in": int(timed_mb.nbytes()) if merge_packed else int(sum(getattr(mb, f).nbytes for f in MultiBatch.FIELDS)),
                 "what": ("%s (solve_merge_region, src/merge_solver.rs:110-200): multi-region batch (%s form) in %s host memory -> pairs + classification on the GPU -> status, "
                          "classification and members in host arrays") % ("avk_merge_packed" if merge_packed else "avk_merge_batch", "packed" if merge_packed else "wide", where)}
        if merge_packed:  # the same call with the wide arrays (avk_merge_batch), same outputs
            merge_multi_batch(ctx, wide_mb, mcfg)
            t0 = time.perf_counter()
            for _ in range(nm):
                wres = merge_multi_batch(ctx, wide_mb, mcfg)
            we = time.perf_counter() - t0
            entry["wide_soa"] = {"value": mb.n_regions * nm / we, "ms_per_step": we / nm * 1e3, "bytes_in": int(sum(getattr(mb, f).nbytes for f in MultiBatch.FIELDS)),
                                 "same_outputs": bool(np.array_equal(wres.status, mres.status) and np.array_equal(wres.classification, mres.classification) and
                                                      np.array_equal(wres.members, mres.members)),
                                 "what": "the same through avk_merge_batch (avk_multi_batch arrays in %s host memory)" % where}
        # parity: oracle pairs + the restated classification (oracle/merge_oracle.py) on a sample, and the pair results of every region
        k = 3
        pb5, cs5 = pair_batch_of(mb), oracle_lib.ContigSet(contigs5)
        (st_o, ex_o), entry["cpu_baseline"] = cpu_figure(lambda: oracle_lib.optimize_pairs(lib, pb5, cs5, 50, threads=cpus), mb.n_regions, "merge regions/s")
        entry["cpu_baseline"]["sample"] += " (the pairwise optimize_sequences of solve_merge_region; the classification is a table lookup beside it)"
        entry["gpu_over_cpu"] = entry["value"] / entry["cpu_baseline"]["value"]
        st_o, ex_o = st_o.reshape(-1, 3), ex_o.reshape(-1, 3)
        bad = []
        want_status, want_cls, want_members = mo.classify_k3_majority(st_o, ex_o)  # the restated rule, vectorised (oracle/merge_oracle.py)
        if not np.array_equal(mres.status, want_status):
            bad.append("status")
        if not np.array_equal(mres.classification, want_cls):
            bad.append("classification")
        if not np.array_equal(mres.members, want_members):
            bad.append("members")
        sample = mb.regions()[:2000] if mb.n_regions <= 200_000 else None  # the dict form is slow: small runs only
        if sample is not None:
            dec = mres.decoded()[:len(sample)]
            for m, reg in enumerate(sample):
                if (st_o[m] != 0).any():
                    continue
                w = mo.classify([len(v) for v in reg["inputs"]], lambda i, j: int(ex_o[m][{(0, 1): 0, (0, 2): 1, (1, 2): 2}[(i, j)]]), False, True, None)
                if dec[m][1] is None or dec[m][1][0] != w[0]:
                    bad.append("restated_rule[%d]" % m)
                    break
        entry["parity"] = "bit-identical (pairs: oracle; classification: restated rule)" if not bad else "MISMATCH:" + ",".join(bad)
        entry["classification_counts"] = {name: int((mres.classification == code).sum()) for name, code in (("different", 0), ("identical", 1), ("majority", 3))}
        if bad:
            print("PARITY FAILURE in the merge leg: %s" % entry["parity"], file=sys.stderr)
            sys.exit(3)
        sec["merge_3_callers"] = entry
        log("secondary merge: %.2f ms per call" % entry["ms_per_step"])
    if not args.no_e2e:
        sec["e2e_compare"] = e2e_leg(args, log)
    return sec


def merge_leg_sharded(ctx, args, rank, world, torch, dist, dev, cpus, log, fence):
    """BASELINE configs[4] on `world` GPUs (SURVEY.md 8e "merge (config 5): same sharding"): ONE merge job of three call sets; merge regions are mapped like compare
    regions (src/main.rs:463-478), so every rank cuts its shard of the packed multi-region batch by avk_region_shard (avk_packed_multi_shard_make), solves it
    with avk_merge_packed, and the job's only cross-region state — MergeSummaryWriter's (reason, type, input) -> (pass, fail) counters
    (src/writers/merge_summary.rs:12-18), a dense block of sums per rank (avk_merge_counts) — is added up with one all-reduce (RCCL).  Timed like the headline:
    barrier + synchronise on both sides, MAX over ranks; value = the job's regions / that time.  Parity: the ranks' results gathered as one checksum and the reduced
    counters against the oracle's pairs + the restated classification on rank 0."""
    import numpy as np
    import oracle_lib
    import merge_oracle as mo
    from aardvark_amd import dist as avk_dist, synth
    from aardvark_amd.merge import MergeConfig, MergeResult, PackedMultiBatch, merge_counts, merge_multi_batch, pinned_multi_batch, shard_packed_multi
    contigs5, mb = synth.config_genome_merge(scale=args.merge_scale, k=3, threads=max(1, min(8, cpus // world)))
    ctx.upload_reference(contigs5)
    whole = PackedMultiBatch.from_multi(mb)
    shard, idx = shard_packed_multi(ctx.lib, whole, mb.region_id, rank, world)
    if not args.pageable:
        shard = pinned_multi_batch(ctx, shard)
    mcfg = MergeConfig(majority_voting_enabled=True)
    mres = merge_multi_batch(ctx, shard, mcfg)
    counts = torch.zeros(int(merge_counts(ctx.lib, shard, mres).size), dtype=torch.int64, device=dev)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    nm = 6
    fence()
    t0 = time.perf_counter()
    for _ in range(nm):
        mres = merge_multi_batch(ctx, shard, mcfg)
    mine = merge_counts(ctx.lib, shard, mres)  # add_merge_benchmark over the rank's regions (the job's batches would add into the same block)
    counts.copy_(torch.from_numpy(mine.view(np.int64).copy()))
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)  # the merge job's only collective
    fence()
    me = time.perf_counter() - t0
    t = torch.tensor([me], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    me = float(t.item())
    job_counts = counts.cpu().numpy().view(np.uint64)
    # one integer for the shards' per-region results: sum over regions of hash(region_id, status, classification, members), summed over the ranks
    def checksum(ids, r):
        with np.errstate(over="ignore"):
            w = (r.status.astype(np.int64).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + r.classification.astype(np.uint64) * np.uint64(1000003) +
                 r.members.astype(np.uint64) * np.uint64(998244353))
            return avk_dist.region_hash(np.asarray(ids, np.uint64) * np.uint64(3) + np.uint64(1) + w * np.uint64(0xD6E8FEB86659FD93)).sum(dtype=np.uint64)
    chk = torch.from_numpy(np.array([checksum(mb.region_id[idx], mres)], np.uint64).view(np.int64).copy()).to(dev)
    dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    sizes = torch.tensor([shard.n_regions], dtype=torch.int64, device=dev)
    gathered = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(gathered, sizes)
    entry = None
    bad = []
    if rank == 0:
        lib = oracle_lib.load()
        st_o, ex_o = oracle_lib.optimize_pairs(lib, pair_batch_of(mb), oracle_lib.ContigSet(contigs5), 50, threads=cpus)
        want = MergeResult(*mo.classify_k3_majority(st_o, ex_o), 3)
        if int(chk.cpu().numpy().view(np.uint64)[0]) != int(checksum(mb.region_id, want)):
            bad.append("job_checksum")
        if not np.array_equal(job_counts, merge_counts(ctx.lib, whole, want)):
            bad.append("summary_counters")
        entry = {"workload": "BASELINE configs[4] stand-in: ONE merge job of 3 call sets (seeds 20250105-7) x %.3g genome, majority strategy, %d regions, sharded by hash(region_id) over %d GPUs"
                             % (args.merge_scale, mb.n_regions, world),
                 "value": mb.n_regions * nm / me, "unit": "merge regions/s", "ms_per_step": me / nm * 1e3, "steps": nm, "n_gpus": world, "scaling": "strong",
                 "regions_per_rank": [int(g.item()) for g in gathered], "summary_counters": int(job_counts.size), "variants_counted": int(job_counts.sum()),
                 "what": "every rank: avk_packed_multi_shard_make -> avk_merge_packed on its shard (host arrays -> host arrays) x %d, then avk_merge_counts and ONE all-reduce of the "
                         "(reason, type, input) -> (pass, fail) sums (src/writers/merge_summary.rs:12-18); max over ranks" % nm,
                 "parity": "bit-identical (checksum of every region's status / classification / members over the ranks, and the reduced summary counters, vs oracle pairs + the restated rule)"
                           if not bad else "MISMATCH:" + ",".join(bad)}
        log("merge sharded over %d ranks: %.2f ms per job" % (world, entry["ms_per_step"]))
    ok = torch.tensor([0 if bad else 1], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        print("PARITY FAILURE in the sharded merge leg on rank %d: %s" % (rank, bad), file=sys.stderr)
        sys.exit(3)
    return entry


def e2e_leg(args, log):
    """The wall-clock half of the metric: the whole `compare` run — FASTA + BED + two VCF.gz of the headline workload's shape written to disk, then the command-line
    tool (aardvark_amd/csrc/cli/compare_main.cpp, the stand-in for the reference's run_compare, src/main.rs:30-327) in fresh processes, the last run checked
    against the oracle (summary.tsv byte for byte, every record of both annotated VCFs).  tools/e2e_genome.py does the work; this parses what it prints."""
    import re
    exe = os.path.join(ROOT, "aardvark_amd", "bin", "aardvark_amd_compare")
    if not os.path.exists(exe):
        return {"skipped": "aardvark_amd/bin/aardvark_amd_compare is not built (make -C aardvark_amd/csrc/cli)"}
    env = dict(os.environ, SCALE=str(args.scale), RUNS="3", VERIFY="1", AVK_TIMING="1")  # AVK_TIMING: the library's own clocks of the tool's one solve, in the last run's lines
    env.pop("AVK_BENCH_CHILD", None)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_genome.py")], capture_output=True, text=True, env=env, timeout=1500)
    except subprocess.TimeoutExpired:
        return {"skipped": "tools/e2e_genome.py did not finish in 1500 s"}
    text = r.stdout
    if r.returncode != 0:
        return {"failed": "tools/e2e_genome.py exit %d" % r.returncode, "stderr_tail": r.stderr[-400:]}
    walls = [float(x) for x in re.findall(r"^exit 0, wall ([0-9.]+) s", text, re.M)]
    tool = [float(x) for x in re.findall(r"Comparisons completed in ([0-9.]+) seconds", text)]
    stages = re.findall(r"stages \[s\]: load ([0-9.]+) .*?regions ([0-9.]+) .*?solve \(pack \+ H2D \+ kernels \+ D2H\) ([0-9.]+), summary \+ annotated VCFs ([0-9.]+)", text)
    loaded = re.search(r"Loaded (\d+) truth and (\d+) query variants; (\d+) regions", text)
    entry = {"workload": "the headline workload as files: 24 contigs x %.3g, FASTA + confident BED + truth.vcf.gz + query.vcf.gz (bgzip), `aardvark_amd_compare -r -t -q -b -o` in a fresh "
                         "process, 3 runs" % args.scale,
             "wall_s": min(tool) if tool else None, "wall_s_runs": tool, "process_wall_s_runs": walls,
             "what": "wall_s = the tool's own clock from its first line to its last (as the reference's, src/main.rs:32,326), best of the runs; process_wall_s includes process start and teardown",
             "stages_s": [{"load": float(a), "regions": float(b), "solve": float(c), "summary_and_vcfs": float(dd)} for a, b, c, dd in stages],
             "regions": int(loaded.group(3)) if loaded else None,
             "summary_identical_to_oracle": "summary.tsv identical to oracle + restated writer: True" in text,
             "per_variant_identical_to_oracle": "per-variant verification PASSED" in text,
             "fixture_seconds": float(re.search(r"written to .* in ([0-9.]+) s", text).group(1)) if re.search(r"written to .* in ([0-9.]+) s", text) else None,
             "leg_seconds": time.perf_counter() - t0}
    # where the first (only) solve of a fresh process spends its time, on the library's own clocks: the last run's AVK_TIMING lines for the whole-genome batch
    up = re.findall(r"avk upload \(device-packed\): (\d+) regions, \d+ calls: buffers ([0-9.]+) ms, copies queued ([0-9.]+) ms, packing kernels \+ plan ([0-9.]+) ms, writers queued ([0-9.]+) ms", text)
    ws = re.findall(r"avk run: workspaces \(([^)]*)\) ([0-9.]+) ms", text)
    cp = re.findall(r"avk compare packed: upload ([0-9.]+) ms, launches ([0-9.]+) ms, download ([0-9.]+) ms", text)
    slow = re.findall(r"avk pool: no cached buffer of (\d+) bytes, hipMalloc ([0-9.]+) ms", text)
    if up and cp:
        u = max(up, key=lambda x: int(x[0]))
        entry["first_call_marks"] = {"regions": int(u[0]), "upload_ms": {"buffers": float(u[1]), "copies_queued": float(u[2]), "packing_kernels_and_plan": float(u[3]), "writers_queued": float(u[4])},
                                     "workspaces": {"what": ws[-1][0], "ms": float(ws[-1][1])} if ws else None,
                                     "call_ms": {"upload": float(cp[-1][0]), "launches": float(cp[-1][1]), "download": float(cp[-1][2])},
                                     "pool_misses": len(slow), "slowest_hipMalloc_ms": max([float(x[1]) for x in slow], default=0.0),
                                     "what": "AVK_TIMING lines of the last run's whole-genome call: a process's first call; avk_ctx_warmup (beside the parsing) has taken the code objects, "
                                             "the workspaces and the largest pool buffers off it"}
    if not (entry["summary_identical_to_oracle"] and entry["per_variant_identical_to_oracle"]):
        print("PARITY FAILURE in the e2e_compare leg:\n" + text[-1500:], file=sys.stderr)
        sys.exit(3)
    log("secondary e2e_compare: wall %s s (tool clock), solve stage %s s" % (entry["wall_s"], [s["solve"] for s in entry["stages_s"]]))
    return entry


def pair_batch_of(mb):
    """the CompareRegion-shaped pair batch of a MultiBatch (avk_merge_batch builds the same on its side): pair (i < j) of region m in lexicographic order"""
    import numpy as np
    from aardvark_amd import RegionBatch
    k, n = mb.n_inputs, mb.n_regions
    pairs = [(i, j) for i in range(k) for j in range(i + 1, k)]
    ppr = len(pairs)
    rep = lambda a: np.repeat(a, ppr)
    off = mb.in_off.reshape(n, k)
    cnt = mb.in_cnt.reshape(n, k)
    t_off = np.stack([off[:, i] for i, _ in pairs], axis=1).reshape(-1)
    t_cnt = np.stack([cnt[:, i] for i, _ in pairs], axis=1).reshape(-1)
    q_off = np.stack([off[:, j] for _, j in pairs], axis=1).reshape(-1)
    q_cnt = np.stack([cnt[:, j] for _, j in pairs], axis=1).reshape(-1)
    return RegionBatch(rep(mb.region_id), rep(mb.contig_idx), rep(mb.start), rep(mb.end), t_off, t_cnt, q_off, q_cnt, mb.var_pos, mb.var_type, mb.var_zyg,
                       mb.var_raw_space, mb.a0_off, mb.a0_len, mb.a1_off, mb.a1_len, mb.allele_bytes)


if __name__ == "__main__":
    main()
