#!/bin/bash
# GPU box: SQ counters of one format's kernel.  tools/pmc_one.sh bc7 4096
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_one/fmt; rm -rf $OUT; mkdir -p $OUT
python tools/fmt_bench.py $1 $2 3 > $OUT/fmt_bench.jsonl
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmc_$1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc2_$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/pmc2_$1.log 2>&1
python tools/summarize_fmt_pmc.py $OUT | head -60
python - <<PY
import csv,glob
f=glob.glob("$OUT/pmc2_$1/*counter_collection.csv")
if f:
    d={}
    for r in csv.DictReader(open(f[0])):
        if "cvttmi" in r["Kernel_Name"]:
            d.setdefault(r["Dispatch_Id"],{})
            d[r["Dispatch_Id"]][r["Counter_Name"]]=d[r["Dispatch_Id"]].get(r["Counter_Name"],0)+float(r["Counter_Value"])
    print(list(d.values())[-1])
PY
