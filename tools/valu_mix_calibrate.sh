#!/bin/bash
# GPU box: which instruction kinds the SQ_INSTS_VALU_* class counters count.  Runs single-instruction launches of
# tools/build/valu_peak (64 x iters instructions of ONE kind per wave) under the counters and prints, per kind, every class
# counter as a fraction of SQ_INSTS_VALU.  profiles/r05/valu_mix_calibration.txt; tools/summarize_fmt_pmc.py prices a
# kernel's mix with it.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/valu_mix_cal; rm -rf $OUT; mkdir -p $OUT
OPS="v_add_f32 v_sub_f32 v_mul_f32 v_fma_f32 v_fmac_f32 v_min_f32 v_max_f32 v_min3_f32 v_med3_f32 v_cmp_lt_f32_vcc v_cndmask_b32_sgpr_indep v_mov_b32 v_add_u32 v_sub_u32 v_and_b32 v_or_b32 v_lshlrev_b32 v_lshrrev_b32 v_min_u32 v_max_i32 v_mul_u32_u24 v_mad_i32_i24 v_mul_lo_u32 v_add3_u32 v_bfe_u32 v_perm_b32 v_dot4_u32_u8 v_cvt_f32_ubyte0 v_cvt_f32_i32 v_cvt_i32_f32 v_rndne_f32 v_rcp_f32 v_sqrt_f32 v_pk_mul_f32 v_pk_add_f32 v_readlane_b32 v_mov_b32_dpp_quad v_add_u32_sdwa v_cmp_lt_u32_sgpr"
for op in $OPS; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/$op -o c -- tools/build/valu_peak --op $op --waves 4 --iters 200 > $OUT/$op.log 2>&1
done
python3 - <<PY
import csv, glob, os
out = "$OUT"
print("%-28s %8s %8s %8s %8s %8s %8s   (fraction of SQ_INSTS_VALU, last dispatch)" % ("instruction", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS", "INT32", "CVT"))
for d in sorted(glob.glob(os.path.join(out, "*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        print("%-28s (no counters)" % os.path.basename(d)); continue
    disp = {}
    for r in csv.DictReader(open(f[0])):
        e = disp.setdefault(r["Dispatch_Id"], {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    c = list(disp.values())[-1]
    t = c.get("SQ_INSTS_VALU", 0.0) or 1.0
    print("%-28s %8.3f %8.3f %8.3f %8.3f %8.3f %8.3f" % (os.path.basename(d), c.get("SQ_INSTS_VALU_ADD_F32", 0) / t, c.get("SQ_INSTS_VALU_MUL_F32", 0) / t,
          c.get("SQ_INSTS_VALU_FMA_F32", 0) / t, c.get("SQ_INSTS_VALU_TRANS_F32", 0) / t, c.get("SQ_INSTS_VALU_INT32", 0) / t, c.get("SQ_INSTS_VALU_CVT", 0) / t))
PY
