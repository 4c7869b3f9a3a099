"""No GPU needed: the three translation units cross-compile for gfx950 and their ISA holds no operand hazard of the kinds hipcc does not pad
inside / behind inline asm (VALU write -> MFMA / permlane / DPP read within 2 wait states, v_readfirstlane -> VMEM base within 5).  The
MFMA case is a measured hardware fact (tools/ubench/cvt_mfma_hazard.hip) that once sat in every bf16 conversion of the register-resident
kernels."""
import importlib.util
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_unpadded_operand_hazards_in_the_isa():
    spec = importlib.util.spec_from_file_location("isa_hazard_scan", os.path.join(REPO, "tools", "checks", "isa_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0


def test_the_scanner_sees_a_planted_hazard(tmp_path):
    spec = importlib.util.spec_from_file_location("isa_hazard_scan", os.path.join(REPO, "tools", "checks", "isa_hazard_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p = tmp_path / "k.s"
    p.write_text("\tv_cvt_pk_bf16_f32 v7, v2, v3\n\tv_mov_b32_e32 v9, v1\n\tv_mfma_f32_16x16x32_bf16 v[12:15], v[20:23], v[4:7], 0\n"
                 "\tv_cvt_pk_bf16_f32 v7, v2, v3\n\ts_nop 1\n\tv_mfma_f32_16x16x32_bf16 v[12:15], v[20:23], v[4:7], 0\n"
                 "\tv_readfirstlane_b32 s6, v8\n\ts_nop 1\n\tglobal_load_lds_dwordx4 v130, s[6:7]\n"
                 # (round 5) an accumulator read by a conversion inside an asm statement three states behind the MFMA; the same behind s_nop 15 is fine,
                 # and so is hipcc's own conversion (no asm markers: hipcc pads it)
                 "\tv_mfma_f32_16x16x32_bf16 v[40:43], v[20:23], v[24:27], 0\n\tv_mov_b32_e32 v9, v1\n\t;;#ASMSTART\n\tv_cvt_pk_bf16_f32 v50, v40, v41\n\ts_nop 0\n\t;;#ASMEND\n"
                 "\tv_mfma_f32_16x16x32_bf16 v[60:63], v[20:23], v[24:27], 0\n\ts_nop 15\n\t;;#ASMSTART\n\tv_cvt_pk_bf16_f32 v51, v60, v61\n\ts_nop 0\n\t;;#ASMEND\n"
                 "\tv_mfma_f32_16x16x32_bf16 v[70:73], v[20:23], v[24:27], 0\n\tv_cvt_pk_bf16_f32 v52, v70, v71\n")
    found = mod.scan(str(p))
    assert sorted(f[0] for f in found) == sorted(["valu->mfma", "sgpr->vmem", "mfma->asm-valu"]), found
