#!/bin/bash
# Round 4: which regions go to the wide kernel — 50 queued whole-genome resident steps per option string (fresh process each), two repeats
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
run() { printf "%-60s " "$1"; for rep in 1 2; do timeout 120 .scratch/first_step_probe /tmp/w100.bin 50 1 25 "$1" 2>&1 | tail -1 | sed 's/.*finished \([0-9.]*\) ms later.*/\1/' | tr '\n' ' '; done; echo "ms for 50 steps"; }
for o in "$@"; do [ "$o" = "-" ] && o=""; run "$o"; done
