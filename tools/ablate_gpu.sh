#!/bin/bash
# per-kernel durations of the ablation builds (AVK_STOP_AFTER=k) under rocprofv3 --kernel-trace
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ablate; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  AVK_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$lib -o a -- python3 $ROOT/tools/gpu_time.py $lib > $OUT/$lib.log 2>&1
  python3 - <<PY
import sqlite3,glob
for f in glob.glob("$OUT/$lib/*.db"):
    c=sqlite3.connect(f).cursor()
    for r in c.execute("select name, grid_x, lds_size, count(*), avg(duration)/1e3 from kernels where name like 'avk_region%' group by name, grid_x, lds_size order by avg(duration) desc"):
        print("$lib", r[0][:24], "grid", r[1], "lds", r[2], "n", r[3], "avg_us %.1f" % r[4])
PY
done
