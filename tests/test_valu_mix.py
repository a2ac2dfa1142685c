"""tools/valu_mix.py (measurement tooling, CPU): the classification of VALU mnemonics into SQ counter classes and issue-pairing
classes follows the calibration (profiles/r05/valu_mix_calibration.txt) and the pair matrix (profiles/r02/valu_peak.json), and the
slot arithmetic of the issue floor is what its docstring says."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("valu_mix", os.path.join(ROOT, "tools", "valu_mix.py"))
vm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(vm)


def test_classification_matches_the_calibration():
    want = {"v_add_f32_e32": ("ADD_F32", "F"), "v_sub_f32_e32": ("ADD_F32", "F"), "v_mul_f32_e32": ("MUL_F32", "F"), "v_pk_mul_f32": ("MUL_F32", "S2"),
            "v_pk_add_f32": ("ADD_F32", "S2"), "v_fma_f32": ("FMA_F32", "S1"), "v_fmac_f32_e32": ("FMA_F32", "S1"), "v_rcp_f32_e32": ("TRANS_F32", "T"),
            "v_sqrt_f32_e32": ("TRANS_F32", "T"), "v_cvt_f32_ubyte0_e32": ("CVT", "S1"), "v_cvt_f32_i32_sdwa": ("CVT", "S2"), "v_add_u32_e32": ("INT32", "I"),
            "v_sub_u32_e32": ("INT32", "I"), "v_mad_i32_i24": ("INT32", "S2"), "v_mul_u32_u24_e32": ("INT32", "S2"), "v_mul_lo_u32": ("INT32", "S1"),
            "v_bfe_u32": ("INT32", "S1"), "v_min_u32_e32": ("INT32", "S1"), "v_dot4_u32_u8": ("INT32", "S2"), "v_cmp_lt_u32_e32": ("INT32", "S1"),
            "v_and_b32_e32": ("OTHER", "I"), "v_or_b32_e32": ("OTHER", "I"), "v_lshrrev_b32_e32": ("OTHER", "I"), "v_lshlrev_b32_e32": ("OTHER", "S1"),
            "v_mov_b32_e32": ("OTHER", "F"), "v_mov_b32_dpp": ("OTHER", "S2"), "v_min_f32_e32": ("OTHER", "S1"), "v_max_f32_e32": ("OTHER", "S1"),
            "v_cmp_lt_f32_e32": ("OTHER", "S1"), "v_cndmask_b32_e32": ("OTHER", "S1"), "v_perm_b32": ("OTHER", "S1"), "v_readlane_b32": ("OTHER", "S2"),
            "v_med3_f32": ("OTHER", "S1"), "v_rndne_f32_e32": ("OTHER", "S1")}
    for op, exp in want.items():
        assert vm.classify(op) == exp, (op, vm.classify(op))


def test_slot_arithmetic():
    # S1 shares with F first, what is left of F pairs with I and itself, S2 and leftover S1 stand alone
    assert vm.slots({"F": 40, "I": 10, "S1": 30, "S2": 20, "T": 0}) == (20.0, 30.0 + 10.0, 0.0)
    assert vm.slots({"F": 10, "I": 20, "S1": 50, "S2": 0, "T": 4}) == (40.0, 10.0 + 10.0, 4.0)
    c = {"SQ_INSTS_VALU": 100.0, "SQ_INSTS_VALU_ADD_F32": 20.0, "SQ_INSTS_VALU_MUL_F32": 20.0, "SQ_INSTS_VALU_FMA_F32": 10.0, "SQ_INSTS_VALU_TRANS_F32": 0.0,
         "SQ_INSTS_VALU_INT32": 20.0, "SQ_INSTS_VALU_CVT": 10.0}
    split = {"ADD_F32": {"F": 1}, "MUL_F32": {"F": 1}, "FMA_F32": {"S1": 1}, "INT32": {"I": 1, "S2": 1}, "CVT": {"S1": 1}, "OTHER": {"I": 1, "S1": 1}}
    r = vm.floor(c, split)
    # F 40, I 10 + 10, S1 10 + 10 + 10, S2 10: 30 S1-F pairs, (10 F + 20 I) / 2 = 15 pairs, 10 alone
    assert {k: v for k, v in r["pairing_classes"].items() if v} == {"F": 0.4, "I": 0.2, "S1": 0.3, "S2": 0.1}
    assert abs(r["issue_floor_cycles_per_inst"] - (45 * vm.PAIRED_SLOT + 10 * vm.SLOT) / 100.0) < 1e-3
    assert r["issue_floor_cycles_per_inst"] > r["two_cycle_peak_cycles_per_inst"]
    assert r["issue_floor_without_S1_F_sharing"] > r["issue_floor_cycles_per_inst"]


def test_static_split_of_the_shipped_library():
    if not os.path.exists(vm.OBJDUMP):
        pytest.skip("llvm-objdump not installed")
    s = vm.static_split()
    k = [n for n in s if n.startswith("cvttmi_bc7_kernel<true, false, false>")]
    assert len(k) == 1
    total = sum(sum(e.values()) for e in s[k[0]].values())
    assert total > 10000 and s[k[0]]["ADD_F32"].get("S2", 0) == 0  # the BC7 kernels use no packed f32 (cvtt_kernel_common.h, v2f)
