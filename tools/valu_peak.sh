#!/bin/bash
# GPU box: VALU issue-rate microbenchmark (tools/valu_peak.hip) + the same launches under SQ counters.
#   gpurun -- bash tools/valu_peak.sh ; python tools/summarize_valu_peak.py gpurun_out/valu_peak profiles/r02/valu_peak.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/valu_peak; rm -rf $OUT; mkdir -p $OUT
tools/build/valu_peak > $OUT/valu_peak.jsonl 2> $OUT/valu_peak.err
rocm-smi --showclocks > $OUT/clocks.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc -o valu -- tools/build/valu_peak --waves 4 --iters 500 > $OUT/pmc.log 2>&1
tail -n 5 $OUT/valu_peak.jsonl
ls $OUT/pmc
