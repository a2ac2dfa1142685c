#!/bin/bash
# Runs on the GPU box (via gpurun): the driver's bench command, then rocprofv3 kernel-trace stats and separate PMC passes of
# THE SAME command (--steps 20 --warmup 5, headline only), so that the per-kernel averages in kernel_stats.csv are averages
# over warm dispatches and can be held against the bench line's ms_per_step.  Output under gpurun_out/<tag>/ ; copy the
# summaries into profiles/.
#   tools/profile_round.sh r03 [extra bench args]
TAG=${1:-r03}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
STEPS="--steps 20 --warmup 5"
python bench.py $STEPS "$@" > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bc7 -- python bench.py $STEPS --no-cpu --no-extra "$@" > $OUT/trace.log 2>&1
# counters in their own runs (never together with --stats or a trace domain other than the kernel trace); the summary takes
# the LAST dispatch of every kernel -- 24 launches after start-up, clocks and caches warm -- and the mean over the timed ones
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o bc7 -- python bench.py $STEPS --no-cpu --no-extra "$@" > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bc7 -- python bench.py $STEPS --no-cpu --no-extra "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bc7 -- python bench.py $STEPS --no-cpu --no-extra "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/pmc_mix -o bc7 -- python bench.py $STEPS --no-cpu --no-extra "$@" > $OUT/pmc_mix.log 2>&1
grep -o '{"metric.*' $OUT/bench.json | cut -c1-1500
cat $OUT/trace/bc7_kernel_stats.csv
python tools/summarize_pmc.py $OUT
