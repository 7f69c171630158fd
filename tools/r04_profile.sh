#!/bin/bash
# Round 4: the rocprofv3 evidence of the round (kernel stats + PMC passes of the short bench command) and its summary into profiles/
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
TAG=${1:-r04}
bash tools/profile_round.sh $TAG 2>&1 | tail -12
cd $R
python3 tools/summarize_round.py $TAG > gpurun_out/${TAG}_summary.txt 2>&1; echo "summary rc $?"
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
find gpurun_out/prof_$TAG -name "*.db" -delete   # (the databases are 100 MB; what is judged are the summaries)
tail -40 gpurun_out/profiles_$TAG/${TAG}_pmc.txt
