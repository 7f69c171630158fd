#!/bin/bash
# Round 3, on the GPU box: fresh-process FIRST host-boundary calls in the packed form (two-stream upload, prefix sums, packing, solve, download), three calls
# per process, until the time is up; alternately a quarter and a whole genome.  A run that does not finish in 90 s or exits non-zero is counted and printed.
# usage: tools/r03_first_calls.sh <seconds for the loop>
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
LOOP_S=${1:-480}
LOG=gpurun_out/r03_first_calls.log
: > $LOG
t0=$(date +%s)
i=0; bad=0
while [ $(( $(date +%s) - t0 )) -lt $LOOP_S ]; do
  if [ $(( i % 3 )) -eq 2 ]; then sc=1.0; else sc=0.25; fi
  out=$(timeout 90 python tools/boundary_once.py $sc 2 2>&1 | tr '\n' ' '); rc=$?
  echo "run $i rc $rc scale $sc: $out" >> $LOG
  case "$out" in *"call 2:"*) ;; *) rc=99;; esac
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i rc $rc: $out"; fi
  i=$((i+1))
done
echo "fresh-process runs of three packed boundary calls: $i, not ok: $bad, $(( $(date +%s) - t0 )) s" | tee -a $LOG
tail -2 $LOG | cut -c1-300
