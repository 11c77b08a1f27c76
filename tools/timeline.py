"""Prints the kernel timeline of the last training step in a rocprofv3 kernel_trace.csv (start, end, duration, gap)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "eeg_filter" in r["Kernel_Name"]]
s = idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
names = ["lstm_fwd_persist", "lstm_bwd_persist", "lstm_cell_bwd_il", "lstm_cell_fwd_il", "gemm_nt", "gemm_tn", "colsum",
         "reduce_slabs", "eeg_filter", "blockify", "cast_strided", "fillBuffer", "copyBuffer", "cosine", "multi_tensor",
         "elementwise", "permute", "transpose", "bias_perm", "upcast", "Cijk", "reduce_kernel"]
def short(n):
    for k in names:
        if k in n:
            return k
    return n[:30]
prev_end = 0.0
tot = {}
busy = 0.0
for r in rows[s:]:
    st = (int(r["Start_Timestamp"]) - t0) / 1e3
    en = (int(r["End_Timestamp"]) - t0) / 1e3
    nm = short(r["Kernel_Name"])
    tot.setdefault(nm, [0, 0.0])
    tot[nm][0] += 1
    tot[nm][1] += en - st
    if len(sys.argv) > 2:
        print(f"{st:9.1f} {en:9.1f} dur {en-st:8.1f} gap {st-prev_end:6.1f} s{r['Stream_Id']} {nm} grid={r['Grid_Size_X']}")
    busy += max(0.0, en - max(st, prev_end))
    prev_end = max(prev_end, en)
print(f"step span {prev_end/1e3:.3f} ms, busy {busy/1e3:.3f} ms")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:20s} calls {n:4d} total {t/1e3:7.3f} ms avg {t/n:8.1f} us")
