#!/bin/bash
# GPU box: instruction-cache counters of one format's kernel.  tools/pmc_icache.sh bc7 4096
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_icache; mkdir -p $OUT; rm -rf $OUT/$1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/$1 -o $1 -- python tools/fmt_bench.py $1 $2 1 > $OUT/$1.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$OUT/$1/*counter_collection.csv")
d={}
for r in csv.DictReader(open(f[0])):
    if "cvttmi" in r["Kernel_Name"]:
        k=(r["Dispatch_Id"], r["Kernel_Name"][:40])
        d.setdefault(k,{})
        d[k][r["Counter_Name"]]=d[k].get(r["Counter_Name"],0)+float(r["Counter_Value"])
for k,v in list(d.items())[-3:]:
    print("$1", k, {a:int(b) for a,b in v.items()})
PY
