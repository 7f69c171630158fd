#!/bin/bash
# On the GPU box: the end-to-end tool with its host pools sized by the cgroup's CPU quota (avk_cpus.h) and by the logical CPUs visible (AVK_CPUS=<that many>), N runs each, interleaved.
# usage: tools/e2e_cpus_ab.sh [runs=5]
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
N=${1:-5}
export TMPDIR=${TMPDIR:-/tmp}
SCALE=1.0 RUNS=1 KEEP=1 python tools/e2e_genome.py > /tmp/e2e_gen.log 2>&1
D=$(grep -o "written to [^ ]*" /tmp/e2e_gen.log | head -1 | cut -d" " -f3)
echo "fixtures in $D; cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), logical CPUs $(nproc --all)"
for i in $(seq 1 $N); do
  for mode in ${MODES:-quota visible}; do   # MODES="quota zlib": libdeflate (looked up at run time) against zlib for the BGZF blocks
    unset AVK_CPUS AVF_ZLIB
    [ $mode = visible ] && export AVK_CPUS=$(nproc --all)
    [ $mode = zlib ] && export AVF_ZLIB=1
    aardvark_amd/bin/aardvark_amd_compare -r $D/genome.fa -t $D/truth.vcf.gz -q $D/query.vcf.gz -b $D/hc.bed -o $D/out --disable-variant-trimming > /tmp/e2e_run.log 2>&1
    echo "run $i $mode: $(grep -o 'stages \[s\].*' /tmp/e2e_run.log | sed 's/(side by side[^)]*)//; s/(beside it[^)]*)//; s/(pack + H2D + kernels + D2H)//') | $(grep -o 'Comparisons completed in [0-9.]* seconds' /tmp/e2e_run.log)"
  done
done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat
rm -rf "$D"
