#!/bin/bash
# rocprofv3 evidence for one tag: kernel-trace stats of the bench command, then PMC passes (each in its own run, --kernel-trace only).
# Usage on the GPU box from the repo root: tools/profile_round.sh <tag> [extra bench flags]; then tools/summarize_round.py <tag> (here or there) writes profiles/<tag>_*.
# Every run is checked: a bench that exits non-zero or prints no JSON line under the profiler fails the script's summary line.
set -u
TAG=${1:-r05}
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
python3 -c "import ctypes; l = ctypes.CDLL('$ROOT/aardvark_amd/libaardvark_amd.so'); l.avk_source_hash.restype = ctypes.c_char_p; print(l.avk_source_hash().decode())" > "$OUT/source_hash.txt"
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --resident-steps 5 --no-supervisor $*"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o "$TAG" -- $SHORT > "$OUT/stats.log" 2>&1
echo "stats rc $? json lines $(grep -c '"metric"' "$OUT/stats.log")"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o "$TAG" -- $SHORT > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i ($set) rc $? json lines $(grep -c '"metric"' "$OUT/pmc$i.log")"
done
