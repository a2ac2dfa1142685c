#!/bin/bash
# GPU box: the instruction-ORDER cases of tools/valu_peak.hip (seq:*) at 2 / 3 / 4 waves per SIMD.
#   gpurun -- bash tools/valu_order.sh   -> gpurun_out/valu_order/valu_order.jsonl
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/valu_order; rm -rf $OUT; mkdir -p $OUT
for op in "seq:32cvt_then_32mul" "seq:8cvt_8mul_x4" "seq:4cvt_4mul_x8" "seq:2cvt_2mul_x16" "seq:1cvt_3mul_x16" "seq:4cvt_12mul_x4" "seq:16cvt_48mul" "seq:pixel_as_compiled" "seq:pixel_interleaved" "seq:dot4_perm_block_then_mul" "seq:dot4_perm_spread_in_mul" "pair:mul_f32+cvt_ub0" "v_mul_f32" "v_cvt_f32_ubyte0"; do
  for w in 2 3 4; do
    tools/build/valu_peak --op "$op" --waves $w >> $OUT/valu_order.jsonl 2>> $OUT/err.txt
  done
done
python3 - <<PY
import json
for l in open("$OUT/valu_order.jsonl"):
    if l.startswith("{"):
        r = json.loads(l)
        print("%-34s W=%d  simd cycles per wave-inst %.3f" % (r["op"], r["waves_per_simd"], r["memtime_ghz"] * 1e9 / r["wave_insts_per_s_per_simd(event)"]))
PY
