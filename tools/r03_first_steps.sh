#!/bin/bash
# Round 3, on the GPU box: the transfer rates the host boundary can count on, then fresh-process first steps until the time is up.
# usage: tools/r03_first_steps.sh <seconds for the loop> [steps per run]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LOOP_S=${1:-600}
STEPS=${2:-3}
.scratch/pcie_probe 512 > gpurun_out/r03_pcie_probe.txt 2>&1
cat gpurun_out/r03_pcie_probe.txt
python tools/dump_workload.py 0.25 /tmp/w025.bin
python tools/dump_workload.py 1.0 /tmp/w100.bin
LOG=gpurun_out/r03_first_steps.log
: > $LOG
t0=$(date +%s)
i=0; bad=0
while [ $(( $(date +%s) - t0 )) -lt $LOOP_S ]; do
  mode=$(( i % 2 ))
  if [ $(( i % 4 )) -ge 2 ]; then w=/tmp/w100.bin; else w=/tmp/w025.bin; fi
  out=$(timeout 90 .scratch/first_step_probe $w $STEPS $mode 15 2>&1); rc=$?
  echo "run $i rc $rc $(basename $w) $out" >> $LOG
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i rc $rc: $out"; fi
  i=$((i+1))
done
echo "first-step runs: $i, not ok: $bad, $(( $(date +%s) - t0 )) s" | tee -a $LOG
tail -3 $LOG
