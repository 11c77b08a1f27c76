#!/usr/bin/env python3
"""Build gate: no hot kernel of libcsn_hip.so may use scratch memory.

    python tools/check_spills.py build/csrc/*.res        (the Makefile runs it after every compile)

Why this is a gate and not a perf note: hipcc (ROCm 7.2) spilled a 4-dword MFMA operand of the H = 512 forward as
"3 dwords to scratch + 1 dword kept in an AGPR (reload reuse)" and restored only the three -- the W_hh fragment came
back with a foreign 4th dword, 0.8 % errors in h, present or absent depending on unrelated source edits (DESIGN.md
section 3.5).  The .res files are the compiler's own -Rpass-analysis=kernel-resource-usage remarks.
ALLOWED is empty: the per-diagonal fallbacks and the rarely used filter orders spilled too and were reworked.  The gate
therefore covers EVERY kernel of libcsn_hip.so.

Round 4 made the rule as precise as the bug: what the compiler gets wrong is a spilled register TUPLE split between scratch
and an AGPR ("Reload Reuse").  A kernel whose only scratch traffic is whole single-dword spills of scalar values
(`scratch_store_dword` / `scratch_load_dword`, "4-byte Folded Spill / Reload", no "Reload Reuse" anywhere in the function) has
nothing to split; it passes with a note when the ISA listing next to the remarks shows exactly that -- and fails otherwise,
including when the listing is missing.  (One kernel is in that state: lstm_fwd_ns_kernel<32, true, false> parks the thread
index across its recurrence loops, 8 bytes per lane, no reload inside a loop; the spill-free allocations of the same source
are 7 % slower per launch)"""
import re
import subprocess
import sys

ALLOWED = ()      # since round 3 every kernel of the library is spill-free; nothing is exempt


def parse(path):
    """{function: {metric: value}} from the compiler's remarks.  Two line formats occur -- `file:line:col: remark:     VGPRs: 256 [...]`
    (plain compile) and `remark: file:line:col:     VGPRs: 256 [...]` (with -save-temps, the Makefile's form) -- so the metric is
    looked for anywhere behind the word "remark"."""
    cur, out = None, {}
    for ln in open(path, errors="replace"):
        if "remark" not in ln:
            continue
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        for key in ("ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "VGPRs", "AGPRs"):
            m = re.search(r"(?<![\w ])\s*" + re.escape(key) + r": (\d+)", ln)
            if m and cur:
                out[cur][key] = int(m.group(1))
                break
    return out


def benign_scalar_spills(res_path, name):
    """True iff the function's ISA (the -save-temps listing next to the .res) has only whole single-dword scratch accesses and no
    'Reload Reuse' copy: the partial-tuple reload bug cannot occur."""
    import os
    lst = res_path[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s"
    if not os.path.exists(lst):
        return False, "no ISA listing"
    text = open(lst, errors="replace").read()
    i = text.find("\n" + name + ":")
    j = text.find(".Lfunc_end", i)
    if i < 0 or j < 0:
        return False, "function not found in the listing"
    body = text[i:j]
    acc = [ln.strip() for ln in body.split("\n") if "scratch_" in ln]
    if "Reload Reuse" in body:
        return False, "'Reload Reuse' copies present"
    wide = [a for a in acc if not re.match(r"scratch_(load|store)_dword\s", a)]
    if wide or not acc:
        return False, f"{len(wide)} scratch accesses wider than one dword" if wide else "no scratch instruction found"
    return True, f"{sum(a.startswith('scratch_store') for a in acc)} single-dword spill(s), {sum(a.startswith('scratch_load') for a in acc)} reload(s)"


def main(paths):
    bad, noted, unread = [], [], []
    for p in paths:
        for name, d in parse(p).items():
            if "ScratchSize [bytes/lane]" not in d or "VGPRs Spill" not in d:
                unread.append((p, name))          # the remark format drifted: a gate that reads nothing must not pass
            if d.get("ScratchSize [bytes/lane]", 0) or d.get("VGPRs Spill", 0):
                ok, why = benign_scalar_spills(p, name)
                (noted if ok or any(a in name for a in ALLOWED) else bad).append((p, name, d, why))
    for p, name in unread[:5]:
        print(f"check_spills: ERROR: no scratch / spill figures found for {name[:70]} in {p} (remark format not understood)", file=sys.stderr)
    for p, name, d, why in noted:
        print(f"check_spills: note: {name[:60]} uses {d.get('ScratchSize [bytes/lane]')} B/lane of scratch: {why}, no tuple involved")
    for p, name, d, why in bad:
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        print(f"check_spills: ERROR: {name} uses scratch ({why}): {d}  ({p})", file=sys.stderr)
    n = sum(len(parse(p)) for p in paths)
    print(f"check_spills: {n} kernels in {len(paths)} file(s), {len(bad)} with scratch" + (f", {len(noted)} with scalar spills only" if noted else "") + (f", {len(unread)} unread" if unread else ""))
    return 1 if (bad or unread) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
