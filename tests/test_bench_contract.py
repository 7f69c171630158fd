"""bench.py's output contract: exactly one line on stdout, a JSON object with the agreed keys — also when a RCCL communicator is up
(RCCL writes a version banner to file descriptor 1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "cpu_baseline", "resident_value", "resident", "wide_soa", "dwfa_byte_compares_per_s", "secondary"]


@pytest.mark.gpu
@pytest.mark.parametrize("force_dist", ["0", "1"])
def test_one_json_line_on_stdout(force_dist):
    env = dict(os.environ, AVK_BENCH_FORCE_DIST=force_dist, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--scale", "0.01", "--resident-steps", "4", "--merge-scale", "0.01", "--secondary-scale", "0.01"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:500]
    out = json.loads(lines[0])
    assert [k for k in KEYS if k not in out] == []
    assert out["metric"] == "compared regions/sec (whole node)" and out["unit"] == "regions/s" and out["n_gpus"] == 1 and out["steps"] == 4 and out["warmup"] == 2
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "u8" and out["data"] == "synthetic"
    assert out["config"]["parity"] == "bit-identical" and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "regions/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert 0 < cb["parallel_efficiency"] < 1.5 and cb["one_thread_value"] > 0
    assert out["resident_value"] > 0 and out["resident"]["steps"] == 4
    assert out["value"] > 0 and out["ms_per_step"] > 0
    sec = out["secondary"]
    assert {"shard_1_of_8", "chr20_snv", "dense_mix", "min_variant_gap_1000", "merge_3_callers"} <= set(sec)
    assert all(sec[k]["parity"].startswith("bit-identical") for k in ("shard_1_of_8", "dense_mix", "min_variant_gap_1000", "merge_3_callers", "chr20_snv"))
    assert all(sec[k]["cpu_baseline"]["value"] > 0 and sec[k]["cpu_baseline"]["cores"] >= 1 for k in ("shard_1_of_8", "dense_mix", "min_variant_gap_1000", "merge_3_callers", "chr20_snv"))
    assert sec["shard_1_of_8"]["strong_scaling_bound"]["boundary"] > 0
    assert out["cpu_baseline"]["all_core_extrapolation"]["value"] >= out["cpu_baseline"]["value"]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_run_the_sharded_job():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), on a box with ONE GPU: both ranks on GPU 0 and the sums carried by gloo
    (AVK_BENCH_ONE_DEVICE / AVK_BENCH_BACKEND; RCCL does not take two ranks on one device).  Everything else is the N-rank path: the compare job cut by hash(region_id),
    the tally all-reduce inside the timed region, the job checksum, and the merge job (configs[4]) cut the same way with its summary counters all-reduced."""
    env = dict(os.environ, AVK_BENCH_BACKEND="gloo", AVK_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29579")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29579",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--scale", "0.01", "--resident-steps", "4", "--merge-scale", "0.01"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[:500]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["parity"] == "bit-identical" and out["value"] > 0
    m = out["secondary"]["merge_3_callers"]
    assert m["n_gpus"] == 2 and m["parity"].startswith("bit-identical") and m["value"] > 0 and len(m["regions_per_rank"]) == 2
    assert sum(m["regions_per_rank"]) > 0 and abs(m["regions_per_rank"][0] - m["regions_per_rank"][1]) < 0.1 * sum(m["regions_per_rank"])
    assert m["variants_counted"] > 0
