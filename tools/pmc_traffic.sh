#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <out.json> [extra bench.py args, e.g. --config cfg4]
# HBM traffic per kernel launch of the bench workload: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (never together with other trace domains), corrected as MI355X_MICROARCH.md "HBM" prescribes for gfx950.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/pmc_traffic.json}
shift
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcb_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcb_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-retrieval --no-f32-line --no-parity "$@" > gpurun_out/pmcb_$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/pmcb_$c.log; exit 1; }
done
python3 - "$out" "$*" <<'PY'
import csv, glob, json, sys, collections
res = {"bench_args": sys.argv[2], "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-retrieval --no-f32-line --no-parity",
       "units": "counter values are KiB per dispatch; corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
       "kernels": {}}
names = ["lstm_bwd_persist_kernel", "lstm_fwd_persist_kernel", "lstm_fwd_ns_kernel", "lstm_fwd_f32_persist_kernel", "lstm_bwd_f32_persist_kernel", "gemm_generic_kernel", "gemm_f32_128_kernel", "gemm_nt_wide_kernel", "gemm_nt_256_kernel", "gemm_nt_dma_kernel", "gemm_nt_bf16_kernel",
         "gemm_tn_256_kernel", "gemm_tn_bf16_kernel", "eeg_filter_scan_kernel", "colsum_partial_kernel", "lstm_cell_bwd_il_kernel"]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmcb_{c}/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        for n in names:
            if n in r["Kernel_Name"]:
                tot[n][0] += float(r["Counter_Value"]); tot[n][1] += 1
                break
    for n, (v, k) in tot.items():
        e = res["kernels"].setdefault(n, {})
        e[f"{c}_KB_avg_per_launch"] = v / k
        e[f"launches_{'fetch' if c == 'FETCH_SIZE' else 'write'}"] = k
for n, e in res["kernels"].items():
    if "FETCH_SIZE_KB_avg_per_launch" in e and "WRITE_SIZE_KB_avg_per_launch" in e:
        e["hbm_bytes_per_launch_corrected"] = (2 * e["FETCH_SIZE_KB_avg_per_launch"] + e["WRITE_SIZE_KB_avg_per_launch"]) * 1024
sys.path.insert(0, ".")
import bench
res["csrc_sha16"] = bench.csrc_sha16()     # bench.py drops these values once the kernel sources differ
json.dump(res, open(sys.argv[1], "w"), indent=1)
for n, e in res["kernels"].items():
    print(n, {k: round(v, 1) if isinstance(v, float) else v for k, v in e.items()})
PY
rm -rf gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE
