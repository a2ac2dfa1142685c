#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of
# `python bench.py`.  Output under gpurun_out/<tag>/ ; copy the summaries into profiles/.
#   tools/profile_round.sh r01 [extra bench args]
TAG=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 5 --warmup 2 "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bc7 -- python bench.py --steps 5 --warmup 2 --no-cpu --no-extra "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o bc7 -- python bench.py --steps 1 --warmup 0 --no-cpu --no-extra "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bc7 -- python bench.py --steps 1 --warmup 0 --no-cpu --no-extra "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bc7 -- python bench.py --steps 1 --warmup 0 --no-cpu --no-extra "$@" > $OUT/pmc_write.log 2>&1
grep -o '{"metric.*' $OUT/bench.json | cut -c1-2000
cat $OUT/trace/bc7_kernel_stats.csv
python tools/summarize_pmc.py $OUT
