#!/bin/bash
# Build container: copy the summaries of a finished tools/profile_round.sh + tools/profile_formats.sh run (merged back under
# gpurun_out/<tag>/) into profiles/<tag>/ -- the files bench.py and the judge read.   tools/install_profiles.sh r04
set -e
TAG=${1:-r04}
SRC=gpurun_out/$TAG
DST=profiles/$TAG
mkdir -p $DST
cp $SRC/summary.json $DST/summary.json
cp $SRC/trace/bc7_kernel_stats.csv $DST/kernel_stats.csv
cp $SRC/fmt/fmt_summary.json $DST/formats_summary.json
cp $SRC/fmt/fmt_bench.jsonl $DST/formats_bench.jsonl
for d in $SRC/fmt/trace_*; do
  f=${d##*/trace_}
  [ -d "$d" ] && [ "$f" != "bc7" ] && cp $d/${f}_kernel_stats.csv $DST/kernel_stats_$f.csv
done
python tools/kernel_resources.py > $DST/kernel_resources.txt
python - <<PY
import json
a = json.load(open("$DST/summary.json")); b = json.load(open("$DST/formats_summary.json"))
print("profiles/$TAG: source_sha256", a.get("source_sha256"), b.get("source_sha256") == a.get("source_sha256"))
PY
