#!/bin/bash
# The whole-genome `compare` run (tools/e2e_genome.py fixtures) with the packed and the wide batch form, the library's stage timing on (AVK_TIMING): what the
# first — and only — call of a fresh process spends where.  usage: tools/e2e_forms.sh  (GPU box; writes gpurun_out/e2e_forms.txt)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for form in packed wide packed wide; do
    echo "== --batch-form $form"
    AVK_TIMING=1 SCALE=1 CLI_ARGS="--batch-form $form" timeout 400 python tools/e2e_genome.py 2>&1 | grep -E "^avk |^stages|^Comparisons|^exit"
done
} > gpurun_out/e2e_forms.txt 2>&1
cat gpurun_out/e2e_forms.txt
