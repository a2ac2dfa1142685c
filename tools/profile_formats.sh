#!/bin/bash
# Runs on the GPU box (via gpurun): per-format kernel timing, then rocprofv3 kernel stats and two PMC passes (SQ issue
# counters; LDS / memory instruction counters) for the non-headline kernels AT THE BASELINE CONFIG SIZES (4096^2).
# Output under gpurun_out/<tag>/fmt/ ; tools/summarize_fmt_pmc.py condenses it into fmt_summary.json.
#   tools/profile_formats.sh r02
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG/fmt
mkdir -p $OUT
: > $OUT/fmt_bench.jsonl
if [ -z "$PROFILE_SKIP_BENCH" ]; then
for spec in "bc7 4096" "bc7o 4096" "bc7b 4096" "bc7u 4096" "bc7photo 4096" "bc7grad 4096" "bc7two 4096" "bc7c5 16384" "bc7c5u 16384" "bc1 4096" "bc1x 2048" "bc2 4096" "bc3 4096" "bc4 4096" "bc5 4096" "bc6hu 4096" "bc6hs 2048" "etc1 2048" "etc2 4096" "etc2rgba 4096" "etc2pt 2048" "eac 4096"; do
  set -- $spec
  python tools/fmt_bench.py $1 $2 3 >> $OUT/fmt_bench.jsonl 2>> $OUT/fmt_bench.err
done
fi
cat $OUT/fmt_bench.jsonl
# PROFILE_FORMATS="bc6hu:4096,bc7o:4096" restricts the counter passes
IFS=',' read -ra SPECS <<< "${PROFILE_FORMATS:-bc7:4096,bc7o:4096,bc7u:4096,bc6hu:4096,etc2rgba:4096,bc1:4096,bc7photo:4096,bc7grad:4096,bc7two:4096,bc7c5:16384,bc7c5u:16384}"
for spec in "${SPECS[@]}"; do
  spec=${spec/:/ }
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1 -o $1 -- python tools/fmt_bench.py $1 $2 3 > $OUT/trace_$1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmc_$1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc2_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmc2_$1.log 2>&1
  # HBM traffic: FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, HBM / rocprofv3 section)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcf_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmcf_$1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcw_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmcw_$1.log 2>&1
  # instruction-class mix (tools/valu_mix.py)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/pmcm_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmcm_$1.log 2>&1
  cat $OUT/trace_$1/*kernel_stats.csv | head -4
done
python tools/summarize_fmt_pmc.py $OUT > /dev/null
python - <<PY
import json
d = json.load(open("$OUT/fmt_summary.json"))
for k, v in d.items():
    if isinstance(v, list) and v and "derived" in v[0]:
        for e in v:
            print(k, e["kernel"][:50], "vgpr", e["vgpr"], "scratch", e["scratch"], "dur_us", e["dur_us"], e.get("derived"), e.get("mem"))
PY
