#!/bin/bash
# On the GPU box: the tool's first (only) solve of a fresh process, N times, with the library's AVK_TIMING lines of the runs whose solve stage is slow.
# usage: tools/e2e_first_solve.sh [runs=10]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
N=${1:-10}
export TMPDIR=${TMPDIR:-/tmp}
SCALE=1.0 RUNS=1 KEEP=1 python tools/e2e_genome.py > /tmp/e2e_gen.log 2>&1
D=$(grep -o "written to [^ ]*" /tmp/e2e_gen.log | head -1 | cut -d" " -f3)
echo "fixtures in $D"
for i in $(seq 1 $N); do
  AVK_TIMING=1 aardvark_amd/bin/aardvark_amd_compare -r $D/genome.fa -t $D/truth.vcf.gz -q $D/query.vcf.gz -b $D/hc.bed -o $D/out --disable-variant-trimming > /tmp/e2e_run.log 2>&1
  S=$(grep -o "solve (pack + H2D + kernels + D2H) [0-9.]*" /tmp/e2e_run.log | awk '{print $NF}')
  echo "run $i: solve $S s"; [ -z "$S" ] && { tail -5 /tmp/e2e_run.log; ls $D | head; S=0; }
  if awk "BEGIN{exit !($S > ${SLOW:-0.1})}"; then grep -E "^avk (upload|run|compare|download|warm)|hipMalloc [0-9]*\.[0-9]* ms|avk copy_in" /tmp/e2e_run.log | awk '!/hipMalloc 0\.0/' | cut -c1-250 | head -40; fi
done
rm -rf "$D"
