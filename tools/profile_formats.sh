#!/bin/bash
# Runs on the GPU box (via gpurun): per-format kernel timing + rocprofv3 kernel stats and SQ
# counters for the non-headline kernels.  Output under gpurun_out/<tag>/fmt/.
#   tools/profile_formats.sh r01
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG/fmt
mkdir -p $OUT
: > $OUT/fmt_bench.jsonl
for spec in "bc7 4096" "bc7o 2048" "bc1 4096" "bc1x 2048" "bc2 4096" "bc3 4096" "bc4 4096" "bc5 4096" "bc6hu 2048" "bc6hs 2048" "etc1 2048" "etc2 4096" "etc2rgba 4096" "etc2pt 2048" "eac 4096"; do
  set -- $spec
  python tools/fmt_bench.py $1 $2 3 >> $OUT/fmt_bench.jsonl 2>> $OUT/fmt_bench.err
done
cat $OUT/fmt_bench.jsonl
for spec in "bc1 4096" "bc6hu 1024" "etc2rgba 2048"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1 -o $1 -- python tools/fmt_bench.py $1 $2 3 > $OUT/trace_$1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmc_$1.log 2>&1
  cat $OUT/trace_$1/*kernel_stats.csv | head -5
done
python tools/summarize_fmt_pmc.py $OUT
