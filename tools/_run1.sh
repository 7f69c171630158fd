timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py 2> gpurun_out/bench_final_err.txt > gpurun_out/bench_r02_final.json; echo bench rc $?
bash tools/profile_r02.sh r02
timeout 600 python bench.py 2> gpurun_out/bench_final_err.txt > gpurun_out/bench_r02_final.json; echo bench rc $?; python -c "
import json; d=json.load(open('gpurun_out/bench_r02_final.json')); print(d['ms_per_step'], d['value'], d['host_boundary']['ms_per_call'], d['host_boundary']['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['kernel_ms'], d['secondary']['ms_per_step'], d['secondary']['queued_ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['parallel_efficiency'], d['dwfa_byte_compares_per_s'])"
SEED_BASE=90000 BUDGET_S=240 timeout 500 python tools/gpu_fuzz.py > gpurun_out/gpu_fuzz_r02_final.txt 2>&1; tail -1 gpurun_out/gpu_fuzz_r02_final.txt
python tools/gpu_hang_probe.py 1.0 20 8 2>&1 | tail -2
