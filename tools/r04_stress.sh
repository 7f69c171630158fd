#!/bin/bash
# Round 4: no step and no first call may hang with the round's launch graph: 4,000 queued whole-genome resident steps with progress output (plain and on torch's stream),
# then fresh-process first boundary calls for four minutes (three calls per process, alternately a quarter and a whole genome)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out
timeout 300 python tools/gpu_stress_steps.py 1.0 20 100 2>&1 | tail -4
timeout 300 python tools/gpu_stress_steps.py 1.0 20 100 torch 2>&1 | tail -3
bash tools/r03_first_calls.sh ${1:-240}
cp gpurun_out/r03_first_calls.log gpurun_out/r04_first_calls.log
