#!/bin/bash
# GPU box: the v_fma_f32 issue probe (tools/valu_fma_probe.hip) -- timings, the shader clock while it runs, SQ counters of the 8-waves-per-SIMD
# launches, and the ISA of the timed loop.   gpurun -- bash tools/valu_fma_probe.sh ; results under gpurun_out/valu_fma/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/valu_fma; rm -rf $OUT; mkdir -p $OUT
# the shader clock under load: sampled while a long probe runs
( tools/build/valu_fma_probe 6000000 8 > /dev/null ) &
sleep 2.5; rocm-smi --showclocks > $OUT/clocks_under_load.txt 2>&1; wait
GHZ=$(grep -i sclk $OUT/clocks_under_load.txt | head -1 | sed 's/.*(\([0-9]*\)Mhz).*/\1/' | awk '{printf "%.3f", $1/1000}')
echo "shader clock under load: $GHZ GHz" | tee $OUT/clock.txt
PROBE_GHZ=${GHZ:-2.4} tools/build/valu_fma_probe 4000 > $OUT/probe.jsonl 2> $OUT/probe.err
cat $OUT/probe.jsonl
for v in 8 18 28 38; do   # variant * 10 + waves per SIMD: the four variants at 8 waves per SIMD
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$v -o p -- tools/build/valu_fma_probe 4000 $v > $OUT/pmc_$v.log 2>&1
done
python3 - <<PY
import csv, glob, json
out = {}
for v in (8, 18, 28, 38):
    f = glob.glob("$OUT/pmc_%d/*counter_collection.csv" % v)
    if not f: continue
    acc = {}
    n = 0
    for r in csv.DictReader(open(f[0])):
        if "probe" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n = max(n, int(r["Dispatch_Id"]))
    out[v] = acc
    print(v, {k: int(x) for k, x in acc.items()})
json.dump(out, open("$OUT/pmc.json", "w"), indent=1)
PY
/opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading tools/build/valu_fma_probe > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -S --cuda-device-only -o $OUT/probe.s tools/valu_fma_probe.hip 2>/dev/null
grep -n "s_cbranch\|v_fma_f32\|v_pk_fma\|v_mul_f32 v8\|s_add_i32\|s_cmp" $OUT/probe.s | head -40 > $OUT/loop_isa.txt
