"""tools/asm_lint.py on hand-written assembly: the two things hipcc does not do for an inline-asm statement (CDNA guide 5.7) are
what conv_board.hip's pipelines rely on NOT happening, so the lint that looks for them in the -save-temps output gets its own
test: (a) a compiler instruction touching the destination of an asm load that no counted wait has covered yet, (b) a VALU
write of an MFMA operand within two wait states of an asm MFMA (the bug that cost the wide split one accumulator register)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lint(tmp_path, body):
    path = tmp_path / "k.s"
    path.write_text("kernel_under_test:\n" + body + "\ts_endpgm\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_lint.py"), str(path), "kernel_under_test:"],
                         capture_output=True, text=True, check=True)
    return out.stdout


ASM = "\t;;#ASMSTART\n{}\n\t;;#ASMEND\n"


def test_clean_pipeline_passes(tmp_path):
    body = (ASM.format("\tds_read_b128 v[4:7], v20 offset:0") + ASM.format("\tds_read_b128 v[8:11], v20 offset:512") +
            ASM.format("\ts_waitcnt lgkmcnt(1)") + "\tv_add_u32_e32 v30, v4, v5\n" +            # v[4:7] has landed: the older of two reads
            ASM.format("\ts_waitcnt lgkmcnt(0)") + "\tv_add_u32_e32 v31, v8, v9\n")
    assert "0 findings; 0 asm MFMAs" in lint(tmp_path, body)


def test_touching_an_lds_read_in_flight_is_reported(tmp_path):
    body = (ASM.format("\tds_read_b128 v[4:7], v20 offset:0") + ASM.format("\tds_read_b128 v[8:11], v20 offset:512") +
            ASM.format("\ts_waitcnt lgkmcnt(1)") + "\tv_mov_b32_e32 v40, v9\n")                  # the younger read may still be out
    out = lint(tmp_path, body)
    assert "1 findings" in out and "LDS read still in flight" in out


def test_touching_an_asm_global_load_in_flight_is_reported(tmp_path):
    body = (ASM.format("\tglobal_load_dwordx4 v[12:15], v[2:3], off") + "\tglobal_load_lds_dwordx4 v[0:1], off\n" +
            ASM.format("\ts_waitcnt vmcnt(1)") + "\tv_mov_b32_e32 v40, v12\n" +                 # landed: only the DMA may still be out
            ASM.format("\tglobal_load_dwordx4 v[16:19], v[2:3], off") + "\tv_mov_b64_e32 v[50:51], v[16:17]\n")   # copied before any wait
    out = lint(tmp_path, body)
    assert "1 findings" in out and "asm global load still in flight" in out


def test_valu_write_straight_in_front_of_an_asm_mfma_is_reported(tmp_path):
    bad = "\tv_accvgpr_mov_b32 a240, a224\n" + ASM.format("\tv_mfma_f32_32x32x16_f16 a[240:255], v[80:83], v[4:7], a[240:255]")
    good = "\tv_accvgpr_mov_b32 a240, a224\n\ts_nop 1\n" + ASM.format("\tv_mfma_f32_32x32x16_f16 a[240:255], v[80:83], v[4:7], a[240:255]")
    assert "1 asm MFMAs behind a fresh VALU write" in lint(tmp_path, bad)
    assert "0 asm MFMAs behind a fresh VALU write" in lint(tmp_path, good)
