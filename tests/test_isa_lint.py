"""The generated code of every kernel source is free of the reload-behind-LDS-DMA pattern (tools/isa_lint.py): a
VGPR-returning load whose first use is guarded only by counted vmcnt waits although LDS-DMA instructions were issued in
between.  CPU only (hipcc -S cross-compiles gfx950)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_no_counted_wait_guards_a_load_across_lds_dma(capsys):
    import isa_lint
    assert isa_lint.main([]) == 0, capsys.readouterr().out


def test_lint_sees_the_pattern(tmp_path):
    """The detector itself: a hand-written listing with the hazard (the round-2 build of tail.hip had exactly this)."""
    import isa_lint
    listing = "\n".join(["_Zk:", "\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_barrier", "\tglobal_load_dword v48, v[2:3], off offset:16",
                         *["\tbuffer_load_dwordx4 v1, s[12:15], s4 offen lds"] * 16, "\ts_waitcnt vmcnt(16)", "\tv_add_u32_e32 v47, s1, v48",
                         "\ts_endpgm"])
    safe = listing.replace("s_waitcnt vmcnt(16)", "s_waitcnt vmcnt(0)")
    orig = isa_lint.asm_of
    try:
        isa_lint.asm_of = lambda src: listing
        assert len(isa_lint.lint("x.hip")) == 1
        isa_lint.asm_of = lambda src: safe
        assert isa_lint.lint("x.hip") == []
        # second rule: the result of an inline-assembly MFMA read too early / late enough / by the compiler's own MFMA
        early = "\n".join(["_Zk:", "\t;;#ASMSTART", "\tv_mfma_f32_32x32x16_f16 v[2:17], v[20:23], v[24:27], v[2:17]", "\t;;#ASMEND",
                           "\tv_mul_f32_e32 v40, v3, v3", "\ts_endpgm"])
        isa_lint.asm_of = lambda src: early
        assert len(isa_lint.lint("x.hip")) == 1
        isa_lint.asm_of = lambda src: early.replace("\tv_mul_f32", "\ts_nop 7\n\ts_nop 3\n\tv_mul_f32")
        assert isa_lint.lint("x.hip") == []
        isa_lint.asm_of = lambda src: early.replace("\t;;#ASMSTART\n", "").replace("\t;;#ASMEND\n", "")
        assert isa_lint.lint("x.hip") == []
        # third rule: an LDS read still outstanding at a ring barrier (round 1's race) / drained first
        racy = "\n".join(["_Zk:", "\tbuffer_load_dwordx4 v1, s[12:15], s4 offen lds", "\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_barrier",
                          "\tds_read_b128 v[4:7], v2", "\ts_waitcnt vmcnt(0)", "\ts_barrier", "\ts_endpgm"])
        isa_lint.asm_of = lambda src: racy
        assert len(isa_lint.lint("x.hip")) == 1
        isa_lint.asm_of = lambda src: racy.replace("\ts_waitcnt vmcnt(0)\n\ts_barrier", "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier")
        assert isa_lint.lint("x.hip") == []
        # fourth rule: scratch in a kernel with LDS-DMA (allowed only on the allow list)
        spilly = "\n".join(["_Zk:", "\tscratch_store_dword off, v1, off", "\tbuffer_load_dwordx4 v1, s[12:15], s4 offen lds",
                            "\ts_waitcnt vmcnt(0)", "\tscratch_load_dword v1, off, off", "\ts_waitcnt vmcnt(0)", "\tv_add_u32_e32 v2, s1, v1",
                            "\ts_endpgm"])
        isa_lint.asm_of = lambda src: spilly
        assert len(isa_lint.lint("x.hip")) == 1
        isa_lint.asm_of = lambda src: spilly.replace("_Zk:", "_Z17layer_tail_kernelILi512EEv:")
        assert isa_lint.lint("x.hip") == []
        # fifth rule: scratch without LDS-DMA is never wrong, and never intended: any access outside a kernel's budget is reported
        quiet = "\n".join(["_Zk:", "\tscratch_store_dword off, v1, off", "\tscratch_load_dword v1, off, off", "\ts_waitcnt vmcnt(0)",
                           "\tv_add_u32_e32 v2, s1, v1", "\ts_endpgm"])
        isa_lint.asm_of = lambda src: quiet
        assert len(isa_lint.lint("x.hip")) == 1
        isa_lint.asm_of = lambda src: quiet.replace("_Zk:", "_Z17layer_tail_kernelILi512EEv:")
        assert isa_lint.lint("x.hip") == []
        isa_lint.asm_of = lambda src: quiet.replace("_Zk:", "_Z17layer_tail_kernelILi512EEv:").replace(
            "\tscratch_load_dword v1, off, off", "\n".join(["\tscratch_load_dword v1, off, off"] * 50))
        assert len(isa_lint.lint("x.hip")) == 1   # (over its budget of 40)
    finally:
        isa_lint.asm_of = orig
