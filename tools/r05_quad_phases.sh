#!/bin/bash
# round 5: where a quad spends its time (timing build), option sweep around the quads, the quad kernel at 2 / 3 / 4 waves per SIMD
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
AVK_LIB=libaardvark_amd_lanetiming.so timeout 600 python tools/gpu_lane_phases.py 1.0 > gpurun_out/r05_quad_phases.txt 2>&1
AVK_OPTS=lane_quad=0 AVK_LIB=libaardvark_amd_lanetiming.so timeout 600 python tools/gpu_lane_phases.py 1.0 >> gpurun_out/r05_quad_phases.txt 2>&1
cat gpurun_out/r05_quad_phases.txt
tools/sweep_options.sh - lane_split_three=1 het_search_min=5 het_search_min=4 "lane_head_stream=1" "lane_head_stream=1,lane_split_three=1" wide_lazy_blocks=1024 wide_lazy_blocks=256 lane_waves_three=6 lane_waves_three=10 - > gpurun_out/r05_sweep2.txt 2>&1
cat gpurun_out/r05_sweep2.txt
for lib in libaardvark_amd_qwpe2.so libaardvark_amd.so libaardvark_amd_qwpe4.so libaardvark_amd.so; do
  printf "%-34s " "$lib"
  AVK_LIB=$lib timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-parity --steps 40 --resident-steps 300 2>&1 | grep -E "timed region|resident leg" | grep -v '"metric"' | sed 's/.*(\([0-9.]* ms per call\).*/\1/; s/.*resident leg: /resident /' | tr '\n' ' '; echo
done | tee gpurun_out/r05_quad_wpe.txt
