"""CPU: the two build gates on synthetic compiler output -- they are correctness gates (DESIGN.md sections 3.5, 3.8), so
their own logic is tested: tools/check_asm_hazards.py on small ISA listings (the round-3 pattern is flagged, the wait
states that cure it are counted as the hardware counts them, control flow is followed), tools/check_spills.py on
resource remarks."""
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_hazards as hz      # noqa: E402
import check_spills as sp           # noqa: E402

MFMA = "v_mfma_f32_16x16x32_bf16 v[30:33], a[32:35], v[66:69], v[30:33]"


def _listing(tmp_path, body, name="k.s"):
    p = tmp_path / name
    p.write_text("\t.text\n_Z6kernelv:\n" + textwrap.dedent(body) + "\ts_endpgm\n.Lfunc_end0:\n")
    return str(p)


def _asm(ins):
    return f"\t;;#ASMSTART\n\t{ins}\n\t;;#ASMEND\n"


def test_round3_pattern_is_flagged_and_wait_states_cure_it(tmp_path):
    bad = _listing(tmp_path, "\tv_accvgpr_write_b32 a34, v218\n\tv_accvgpr_write_b32 a35, v219\n" + _asm(MFMA), "bad.s")
    found = hz.check(bad)
    assert len(found) == 2 and all(f[3] == "R1" for f in found)
    assert "0 wait state(s) ahead" in found[0][4] or "0 wait state(s) ahead" in found[1][4]
    # one instruction in between is ONE wait state: a35 is still too close, a34 is not
    one = _listing(tmp_path, "\tv_accvgpr_write_b32 a34, v218\n\tv_accvgpr_write_b32 a35, v219\n\ts_nop 0\n" + _asm(MFMA), "one.s")
    assert [f[4].split("`")[1] for f in hz.check(one)] == ["v_accvgpr_write_b32 a35, v219"]
    # s_nop 1 = two wait states (inside or outside the asm block): clean
    for k, body in enumerate(("\tv_accvgpr_write_b32 a35, v219\n\ts_nop 1\n" + _asm(MFMA),
                              "\tv_accvgpr_write_b32 a35, v219\n\t;;#ASMSTART\n\ts_nop 1\n\t" + MFMA + "\n\t;;#ASMEND\n",
                              "\tv_accvgpr_write_b32 a35, v219\n\tv_add_u32_e32 v1, v2, v3\n\tv_add_u32_e32 v4, v2, v3\n" + _asm(MFMA))):
        assert hz.check(_listing(tmp_path, body, f"ok{k}.s")) == []
    # a write of a register the MFMA does not read, or an MFMA the compiler can see (not in an asm block): not this gate's business
    assert hz.check(_listing(tmp_path, "\tv_accvgpr_write_b32 a36, v219\n" + _asm(MFMA), "other.s")) == []
    assert hz.check(_listing(tmp_path, "\tv_accvgpr_write_b32 a35, v219\n\t" + MFMA + "\n", "builtin.s")) == []


def test_hazards_are_followed_through_branches_and_not_through_dead_fallthrough(tmp_path):
    # the writer sits in a block that BRANCHES to the MFMA's block: still a predecessor
    through = _listing(tmp_path, "\tv_mov_b32_e32 v66, 0\n\ts_branch .LBB0_2\n.LBB0_1:\n\tv_mov_b32_e32 v67, 0\n\ts_endpgm\n.LBB0_2:\n" + _asm(MFMA),
                       "through.s")
    f = hz.check(through)
    assert len(f) == 1 and "v_mov_b32_e32 v66, 0" in f[0][4]          # v67's writer is not on a path into the MFMA
    # R2: the result touched too early by a VALU instruction; an MFMA accumulating into the same registers is exempt
    early = _listing(tmp_path, _asm(MFMA) + "\tv_add_f32_e32 v1, v30, v2\n", "early.s")
    assert [x[3] for x in hz.check(early)] == ["R2"]
    chain = _listing(tmp_path, _asm(MFMA) + _asm(MFMA) + "\ts_nop 15\n\tv_add_f32_e32 v1, v30, v2\n", "chain.s")
    assert hz.check(chain) == []
    # ... and not reported when the consumer sits behind an unconditional branch away from it
    away = _listing(tmp_path, _asm(MFMA) + "\ts_branch .LBB0_9\n\tv_add_f32_e32 v1, v30, v2\n.LBB0_9:\n\ts_nop 15\n", "away.s")
    assert hz.check(away) == []
    assert hz.main([through]) == 1 and hz.main([chain]) == 0


def test_spill_gate_reads_the_compilers_remarks(tmp_path):
    res = tmp_path / "x.res"
    res.write_text(textwrap.dedent("""\
        x.hip:1:0: remark: Function Name: _Z4goodv [-Rpass-analysis=kernel-resource-usage]
        x.hip:1:0: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
        x.hip:1:0: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
        x.hip:1:0: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
        x.hip:9:0: remark: Function Name: _Z3badv [-Rpass-analysis=kernel-resource-usage]
        x.hip:9:0: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
        x.hip:9:0: remark:     ScratchSize [bytes/lane]: 28 [-Rpass-analysis=kernel-resource-usage]
        x.hip:9:0: remark:     VGPRs Spill: 10 [-Rpass-analysis=kernel-resource-usage]
        """))
    parsed = sp.parse(str(res))
    assert parsed["_Z4goodv"]["ScratchSize [bytes/lane]"] == 0 and parsed["_Z3badv"]["VGPRs Spill"] == 10
    assert sp.main([str(res)]) == 1
    good = tmp_path / "g.res"
    good.write_text("\n".join(res.read_text().splitlines()[:4]) + "\n")
    assert sp.main([str(good)]) == 0


def _res(tmp_path, name, fn, scratch, spill, fmt):
    rows = [f"Function Name: {fn}", "    VGPRs: 256", f"    ScratchSize [bytes/lane]: {scratch}", f"    VGPRs Spill: {spill}"]
    tag = " [-Rpass-analysis=kernel-resource-usage]"
    text = "".join((f"x.hip:3:0: remark: {r}{tag}\n" if fmt == "plain" else f"remark: x.hip:3:0: {r}{tag}\n") for r in rows)
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_spill_gate_reads_both_remark_formats_and_refuses_what_it_cannot_read(tmp_path):
    # (-save-temps, the Makefile's form since round 4, moves the word "remark" in front of the location: the round-3 parser
    #  matched nothing in that form and passed every build -- the gate had been off for part of round 4)
    for fmt in ("plain", "savetemps"):
        assert sp.main([_res(tmp_path, f"ok_{fmt}.res", "_Z1av", 0, 0, fmt)]) == 0
        assert sp.main([_res(tmp_path, f"bad_{fmt}.res", "_Z1bv", 488, 122, fmt)]) == 1
        assert sp.parse(_res(tmp_path, f"p_{fmt}.res", "_Z1cv", 8, 2, fmt))["_Z1cv"] == {"VGPRs": 256, "ScratchSize [bytes/lane]": 8, "VGPRs Spill": 2}
    # a function whose figures are missing (a format this parser does not know) fails the build instead of passing it
    unread = tmp_path / "u.res"
    unread.write_text("remark: x.hip:3:0: Function Name: _Z1dv [-Rpass-analysis=kernel-resource-usage]\nremark: x.hip:3:0:     Scratch bytes per lane = 0\n")
    assert sp.main([str(unread)]) == 1


def test_spill_gate_passes_whole_scalar_spills_only(tmp_path):
    """The compiler bug the gate exists for is a spilled TUPLE split between scratch and an AGPR ('Reload Reuse').  Whole
    single-dword spills pass with a note -- but only when the ISA listing next to the remarks proves that this is all there is."""
    fn = "_Z1kv"

    def case(name, isa):
        res = _res(tmp_path, name + ".res", fn, 8, 1, "savetemps")
        if isa is not None:
            (tmp_path / (name + "-hip-amdgcn-amd-amdhsa-gfx950.s")).write_text("\t.text\n" + fn + ":\n" + textwrap.dedent(isa) + "\ts_endpgm\n.Lfunc_end0:\n")
        return sp.main([res])

    scalar = "\tscratch_store_dword off, v69, off       ; 4-byte Folded Spill\n\tscratch_load_dword v69, off, off        ; 4-byte Folded Reload\n"
    assert case("scalar", scalar) == 0
    assert case("nolisting", None) == 1
    assert case("tuple", scalar + "\tscratch_store_dwordx3 off, v[38:40], off ; 12-byte Folded Spill\n") == 1
    assert case("reuse", scalar + "\tv_accvgpr_write_b32 a225, v41 ; Reload Reuse\n") == 1
    assert case("wide_load", scalar + "\tscratch_load_dwordx4 v[4:7], off, off offset:16\n") == 1
