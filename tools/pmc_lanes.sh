#!/bin/bash
# GPU box: lane utilisation of one format's kernel (thread-cycles of VALU work against issued VALU instructions).  tools/pmc_lanes.sh bc7 4096
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lanes; mkdir -p $OUT; rm -rf $OUT/$1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/$1.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/$1/*counter_collection.csv")
d={}
for r in csv.DictReader(open(f[0])):
    if "cvttmi" in r["Kernel_Name"]:
        k=(r["Dispatch_Id"], r["Kernel_Name"][:44])
        d.setdefault(k,{})
        d[k][r["Counter_Name"]]=d[k].get(r["Counter_Name"],0)+float(r["Counter_Value"])
best=max(d.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU",0))
k,v=best
print("$1", k[1], {a:int(b) for a,b in v.items()}, "thread_cycles / (insts x 64) = %.3f" % (v["SQ_THREAD_CYCLES_VALU"]/(v["SQ_INSTS_VALU"]*64.0)))
PY
