#!/bin/bash
# Round 4: instruction counters of the wide kernel's launches (one PMC pass, the probe's 3 resident whole-genome steps)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
[ -f /tmp/w100.bin ] || python tools/dump_workload.py 1.0 /tmp/w100.bin > /dev/null
OUT=$R/gpurun_out/wide_pmc
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp AVK_PROBE_TEARDOWN=1 && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o w -- $R/.scratch/first_step_probe /tmp/w100.bin 3 1 25 "${1:-}" > $OUT/p$i.log 2>&1 )
done
python3 - <<PY
import sqlite3, glob, collections
for db in sorted(glob.glob("$OUT/p*/*/w_results.db") + glob.glob("$OUT/p*/w_results.db")):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc_t = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info_t = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    disp_t = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym_t = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = "select s.kernel_name, d.grid_size_x, i.name, sum(p.value), count(distinct d.id) from %s p join %s i on p.pmc_id = i.id join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id where s.kernel_name like 'avk_wide%%' group by s.kernel_name, d.grid_size_x, i.name" % (pmc_t, info_t, disp_t, sym_t)
    for row in cur.execute(q):
        print("%-28s grid %-7d %-24s %14.0f per launch (%d launches)" % (row[0][:28], row[1], row[2], row[3] / row[4], row[4]))
PY
