#!/usr/bin/env python3
"""Build gate: no data hazard around an inline-asm MFMA.

    python tools/check_asm_hazards.py build/csrc/*.s        (the Makefile runs it after every compile)

hipcc's hazard recogniser inserts the wait states gfx950 needs between a VALU write of a register and an MFMA that
reads it (2: `v_add_u32 v3, ...; s_nop 0 / any instruction; s_nop 0; v_mfma ... v[0:3]`) -- but only for MFMAs it can
see.  The N-split forward kernels issue their MFMAs from inline asm (stationary operand constrained to the accumulator
file, lstm_ns_util.h:ns_mfma), and the recogniser does not look inside an asm block.  When register pressure makes the
compiler keep part of a stationary operand in VGPRs and copy it into AGPRs in front of its use, the result is

    v_accvgpr_write_b32 a35, v219
    ;;#ASMSTART
    v_mfma_f32_16x16x32_bf16 v[30:33], a[32:35], v[66:69], v[30:33]       <- reads a35 zero wait states after its write

and the FIRST MFMA behind such a copy multiplies a fragment whose last dword(s) are stale: row group 0 of every tile
differs from row groups 1..3, which reuse the registers a few instructions later.  That is the "wrong row groups" event
of round 3 (fused layer-0 instantiation at H = 1024): found here from the ISA, reproduced and
fixed in round 4.  Whether a build has the pattern depends on register allocation, i.e. on unrelated edits, so it is a
build gate like tools/check_spills.py.

Rules (wait states counted as LLVM's recogniser does: one per instruction issued in between, `s_nop N` = N + 1):
  R1  a VALU instruction (v_*, incl. v_accvgpr_write / v_mov) that writes a register read by an inline-asm MFMA
      (A, B or C operand) must be >= 2 wait states ahead of it;
  R2  a register written by an inline-asm MFMA must not be read or overwritten by a NON-MFMA instruction (compiler- or
      asm-placed) within 18 wait states (16x16x32: 8 passes; 4 + 2 * 8 - ... the recogniser's own number for a
      VALU read/write behind an 8-pass XDL write on gfx940+ is 11; 18 is what lstm_ns_util.h:ns_mfma_fence budgets).
      An MFMA that accumulates into the same registers (C = D) is interlocked by the hardware and exempt.
"""
import re
import sys

R1_WAIT = 2
R2_WAIT = 11

REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+))(?![\w\[])")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        f = m.group(1)
        if m.group(4) is not None:
            out.add((f, int(m.group(4))))
        else:
            out.update((f, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(rest):
    return [o.strip() for o in rest.split(",")] if rest.strip() else []


class Ins:
    __slots__ = ("line", "mn", "ops", "in_asm", "text", "labels")

    def __init__(self, line, mn, ops, in_asm, text, labels=()):
        self.line, self.mn, self.ops, self.in_asm, self.text, self.labels = line, mn, ops, in_asm, text, tuple(labels)

    def is_branch(self):
        return self.mn == "s_branch" or self.mn.startswith("s_cbranch")

    def ends_flow(self):
        return self.mn in ("s_branch", "s_endpgm", "s_setpc_b64")

    def wait_states(self):
        if self.mn == "s_nop":
            try:
                return int(self.ops[0], 0) + 1
            except (ValueError, IndexError):
                return 1
        return 1

    def is_mfma(self):
        return self.mn.startswith("v_mfma") or self.mn.startswith("v_smfmac")

    def is_valu(self):
        return self.mn.startswith("v_") and not self.is_mfma()

    def writes(self):
        if not self.ops:
            return set()
        if self.mn.startswith(("v_cmp", "v_readfirstlane", "v_readlane", "v_cmpx")):
            return set()
        if self.mn.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load", "ds_bpermute", "ds_swizzle")):
            if self.mn.startswith("buffer_load") and "lds" in self.ops[-1]:
                return set()
            return regs(self.ops[0])
        return set()

    def reads(self):
        if self.is_mfma():
            return regs(",".join(self.ops[1:]))
        if self.mn.startswith(("global_store", "buffer_store", "ds_write", "scratch_store")):
            return regs(",".join(self.ops))
        return regs(",".join(self.ops[1:]))


def parse(path):
    """-> {function: [Ins]}"""
    funcs, cur, name, in_asm, pend = {}, None, None, False, []
    for n, raw in enumerate(open(path, errors="replace"), 1):
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";"):
            continue
        m = re.match(r"^([.A-Za-z_][\w.$]*):", raw)
        if m and not raw[0].isspace():
            if not m.group(1).startswith(".L"):
                name = m.group(1)
                cur = funcs.setdefault(name, [])
                pend = []
            else:
                pend.append(m.group(1))
            continue
        if s.startswith(".") or cur is None:
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        cur.append(Ins(n, parts[0], split_ops(parts[1]) if len(parts) > 1 else [], in_asm, s, pend))
        pend = []
    return funcs


def flow(ins):
    """successor / predecessor instruction indices (fall-through + branch targets inside the function)"""
    at = {lb: i for i, x in enumerate(ins) for lb in x.labels}
    succ = [[] for _ in ins]
    for i, x in enumerate(ins):
        if not x.ends_flow() and i + 1 < len(ins):
            succ[i].append(i + 1)
        if x.is_branch() and x.ops and x.ops[0] in at:
            succ[i].append(at[x.ops[0]])
    pred = [[] for _ in ins]
    for i, ss in enumerate(succ):
        for j in ss:
            pred[j].append(i)
    return succ, pred


def check(path):
    bad = []
    for fn, ins in parse(path).items():
        succ, pred = flow(ins)
        for i, m in enumerate(ins):
            if not (m.is_mfma() and m.in_asm):
                continue
            rd = m.reads()
            # R1: VALU writers of an operand fewer than R1_WAIT wait states ahead, on any path into the MFMA
            stack, seen = [(j, 0) for j in pred[i]], set()
            while stack:
                j, ws = stack.pop()
                if ws >= R1_WAIT or (j, ws) in seen:
                    continue
                seen.add((j, ws))
                p = ins[j]
                if p.is_valu() and p.writes() & rd:
                    bad.append((path, fn, m.line, "R1", f"`{p.text}` (line {p.line}) writes an operand of `{m.text}` {ws} wait state(s) ahead (need {R1_WAIT})"))
                stack += [(k, ws + p.wait_states()) for k in pred[j]]
            # R2: non-MFMA users of the result fewer than R2_WAIT wait states behind, on any path out of the MFMA
            stack, seen = [(j, 0, frozenset(regs(m.ops[0]))) for j in succ[i]], set()
            while stack:
                j, ws, wr = stack.pop()
                if ws >= R2_WAIT or not wr or (j, ws, wr) in seen:
                    continue
                seen.add((j, ws, wr))
                q = ins[j]
                if q.is_mfma():
                    d = regs(q.ops[0])
                    if d & wr and not d >= wr and not d <= wr:
                        bad.append((path, fn, q.line, "R2", f"`{q.text}` partially overlaps the result of `{m.text}` (line {m.line})"))
                    wr = wr - d                     # (a later MFMA into the same registers takes over the obligation)
                elif (q.reads() | q.writes()) & wr:
                    bad.append((path, fn, q.line, "R2", f"`{q.text}` touches the result of `{m.text}` (line {m.line}) after {ws} wait state(s) (need {R2_WAIT})"))
                    continue
                stack += [(k, ws + q.wait_states(), wr) for k in succ[j]]
    return sorted(set(bad), key=lambda b: (b[0], b[2]))


def main(paths):
    bad = []
    n_mfma = 0
    for p in paths:
        bad += check(p)
        n_mfma += sum(1 for ins in parse(p).values() for m in ins if m.is_mfma() and m.in_asm)
    for path, fn, line, rule, msg in bad[:40]:
        print(f"check_asm_hazards: ERROR [{rule}] {path}:{line} in {fn[:80]}: {msg}", file=sys.stderr)
    if len(bad) > 40:
        print(f"check_asm_hazards: ... and {len(bad) - 40} more", file=sys.stderr)
    print(f"check_asm_hazards: {n_mfma} inline-asm MFMAs in {len(paths)} file(s), {len(bad)} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
