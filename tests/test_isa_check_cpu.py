"""The build-time guard of vit_ws_gemm.hip's hand-counted panel fetch (maavss_amd/csrc/check_ws_gemm_isa.py) must actually
reject the hazards it exists for: a compiler copy / spill of the fetch registers between the inline-asm loads and their
s_waitcnt, and scratch use.  Synthetic assembly, no GPU and no compiler needed."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_ws_gemm_isa", os.path.join(ROOT, "maavss_amd", "csrc", "check_ws_gemm_isa.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

FETCH = """\t;;#ASMSTART
\tglobal_load_dwordx4 v[112:115], v[8:9], off
\tglobal_load_dwordx4 v[116:119], v[8:9], off offset:1024
\t;;#ASMEND
"""
WAIT = """\t;;#ASMSTART
\ts_waitcnt vmcnt(4)
\t;;#ASMEND
"""


WAIT0 = WAIT.replace("vmcnt(4)", "vmcnt(0)")
# one half-panel step: MFMA + two output stores, executed twice (inner loop) between the fetch and its vmcnt(4) wait
HALF_STEPS = (".LBB0_2:\n\tv_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]\n\tglobal_store_dwordx4 v[30:31], v[40:43], off\n"
              "\tglobal_store_dwordx4 v[30:31], v[44:47], off offset:32\n\ts_cbranch_scc0 .LBB0_2\n")


def kernel(between_loop_fetch_and_wait="", extra="", steps=HALF_STEPS):
    return ("_Z18vit_ws_gemm_kernelILi0ELi0ELi2EEv6WsArgs: ; @kernel\n" + FETCH + "\tv_add_u32_e32 v1, v2, v3\n" + WAIT0 +
            "\tds_write_b128 v5, v[112:115]\n.LBB0_1:\n\ts_barrier\n" + FETCH + steps + between_loop_fetch_and_wait + WAIT +
            "\tds_write_b128 v5, v[112:115]\n" + extra + "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")


def run(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return chk.main(str(p))


def test_clean_kernel_passes(tmp_path):
    assert run(tmp_path, kernel()) == 0


def test_copy_of_a_pending_register_is_rejected(tmp_path):
    assert run(tmp_path, kernel("\tv_mov_b64_e32 v[0:1], v[112:113]\n")) == 1
    assert run(tmp_path, kernel("\tv_add_f32_e32 v117, v117, v2\n")) == 1


def test_spill_is_rejected(tmp_path):
    """scratch traffic while a fetch is in flight is rejected whatever register it names; a spill / reload outside the fetch spans (a
    loop-invariant value parked by the allocator: round 4's LayerNorm-after-the-product variant has one) is tolerated"""
    assert run(tmp_path, kernel("\tscratch_store_dwordx4 off, v[40:43], off\n")) == 1
    assert run(tmp_path, kernel("\tscratch_load_dword v124, off, off\n")) == 1
    assert run(tmp_path, kernel(extra="\tscratch_store_dwordx4 off, v[40:43], off\n")) == 0


def test_fetch_without_wait_is_rejected(tmp_path):
    text = kernel().replace(WAIT + "\tds_write_b128 v5, v[112:115]\n\ts_cbranch", "\ts_cbranch")
    assert run(tmp_path, text) == 1


def test_wait_that_counts_more_memory_operations_than_were_issued_is_rejected(tmp_path):
    """vmcnt(4) after a fetch is only a wait FOR the fetch if four younger vector-memory operations exist"""
    one_store = HALF_STEPS.replace("\tglobal_store_dwordx4 v[30:31], v[44:47], off offset:32\n", "")
    assert run(tmp_path, kernel(steps=one_store)) == 1                    # 2 x 1 store < 4
    straight = HALF_STEPS.replace(".LBB0_2:\n", "").replace("\ts_cbranch_scc0 .LBB0_2\n", "")
    assert run(tmp_path, kernel(steps=straight)) == 1                     # the same two stores, not in a loop: 2 < 4
    assert run(tmp_path, kernel(steps=straight + straight)) == 0          # unrolled: 4


def test_copy_behind_a_non_last_alternative_fetch_block_is_rejected(tmp_path):
    """Round 4: the loop body had three ALTERNATIVE fetch blocks (no statistics / three-part / two-part statistics), laid out one after
    the other; hipcc placed `v_mov v139, v127` right behind the block that loads v[126:127] -- before the load had landed -- and the
    check, which only remembered the registers of the LAST fetch block in layout order, let it pass (the parts = 3 GPU test caught it).
    Until the wait, the registers of every alternative count as pending."""
    alt_a = FETCH.replace("global_load_dwordx4 v[116:119], v[8:9], off offset:1024",
                          "global_load_dwordx4 v[116:119], v[8:9], off offset:1024\n\tglobal_load_dwordx2 v[126:127], v[10:11], off offset:16")
    copy = "\tv_mov_b32_e32 v139, v127\n"
    both = alt_a + copy + "\ts_cbranch_vccnz .LBB0_9\n" + FETCH + ".LBB0_9:\n"
    text = ("_Z18vit_ws_gemm_kernelILi0ELi2ELi2ELi16EEv6WsArgs: ; @kernel\n" + FETCH + WAIT0 + "\tds_write_b128 v5, v[112:115]\n.LBB0_1:\n\ts_barrier\n" +
            both + HALF_STEPS + WAIT + "\tds_write_b128 v5, v[112:115]\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    assert run(tmp_path, text) == 1
    assert run(tmp_path, text.replace(copy, "")) == 0
