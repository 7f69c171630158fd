#!/bin/bash
# On the GPU box: one run of the end-to-end tool with the feeder's and the library's own stage lines (AVF_TIMING=1, AVK_TIMING=1), after a warm-up run.
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
export TMPDIR=${TMPDIR:-/tmp}
SCALE=1.0 RUNS=1 KEEP=1 python tools/e2e_genome.py > /tmp/e2e_gen.log 2>&1
D=$(grep -o "written to [^ ]*" /tmp/e2e_gen.log | head -1 | cut -d" " -f3)
for i in 1 2; do
  AVF_TIMING=1 aardvark_amd/bin/aardvark_amd_compare -r $D/genome.fa -t $D/truth.vcf.gz -q $D/query.vcf.gz -b $D/hc.bed -o $D/out --disable-variant-trimming > /tmp/e2e_run.log 2>&1
done
grep -E "^\[avf\]|stages|Comparisons" /tmp/e2e_run.log | cut -c1-400
ls -la $D/out | head; rm -rf "$D"
