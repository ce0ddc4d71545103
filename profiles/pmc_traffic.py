"""Post-process the two rocprofv3 PMC passes into profiles/rNN_pmc_traffic.json (HBM-side bytes per GEMM launch / per step).

On the GPU box (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one TCC pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots'):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile
  python3 $R/profiles/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write 9 > $R/gpurun_out/pmc_traffic.json
Corrections (MI355X_MICROARCH.md 'HBM'): the counters are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced
reads -> doubled; WRITE_SIZE is exact for 16-byte streaming stores and float atomics.  Eager mode, because counter collection
serialises dispatches anyway and graph replays are not attributed per kernel."""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    per_kernel, total, n = {}, 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                v = float(row["Counter_Value"]) * 1024.0          # KB -> bytes
                k = row["Kernel_Name"]
                a = per_kernel.setdefault(k, [0.0, 0])
                a[0] += v
                a[1] += 1
                total += v
                n += 1
    return per_kernel, total, n


def main():
    fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])      # steps = warmup + timed steps of the profiled run
    fk, ftot, _ = load(fetch_dir, "FETCH_SIZE")
    wk, wtot, _ = load(write_dir, "WRITE_SIZE")
    is_gemm = lambda k: "gemm_kernel" in k or "gemm_xcd_kernel" in k or "gemm_grouped_kernel" in k or "gemm_dw_batch_kernel" in k or "gemm_kg_kernel" in k
    gf = sum(v[0] for k, v in fk.items() if is_gemm(k)) * 2.0       # gfx950: FETCH_SIZE counts half of wide reads
    gn = sum(v[1] for k, v in fk.items() if is_gemm(k))
    gw = sum(v[0] for k, v in wk.items() if is_gemm(k))
    gnw = sum(v[1] for k, v in wk.items() if is_gemm(k))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --mode eager; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); unit KB->bytes",
           "gemm_launches": gn, "gemm_fetch_bytes_per_launch": gf / max(gn, 1), "gemm_write_bytes_per_launch": gw / max(gnw, 1),
           "gemm_traffic_bytes_per_launch": gf / max(gn, 1) + gw / max(gnw, 1),
           "step_fetch_bytes": ftot * 2.0 / steps, "step_write_bytes": wtot / steps,
           "dw_batch_fetch_bytes_per_launch": sum(v[0] for k, v in fk.items() if "gemm_dw_batch_kernel" in k) * 2.0 / max(sum(v[1] for k, v in fk.items() if "gemm_dw_batch_kernel" in k), 1),
           "dw_batch_write_bytes_per_launch": sum(v[0] for k, v in wk.items() if "gemm_dw_batch_kernel" in k) / max(sum(v[1] for k, v in wk.items() if "gemm_dw_batch_kernel" in k), 1),
           "chain_fetch_bytes_per_launch": sum(v[0] for k, v in fk.items() if "chain_fwd_kernel" in k or "chain64_fwd_kernel" in k) * 2.0 / max(sum(v[1] for k, v in fk.items() if "chain_fwd_kernel" in k or "chain64_fwd_kernel" in k), 1),
           "chain_write_bytes_per_launch": sum(v[0] for k, v in wk.items() if "chain_fwd_kernel" in k or "chain64_fwd_kernel" in k) / max(sum(v[1] for k, v in wk.items() if "chain_fwd_kernel" in k or "chain64_fwd_kernel" in k), 1),
           "top_fetch_kernels_bytes_per_launch": {k[:60]: round(v[0] * 2.0 / v[1]) for k, v in sorted(fk.items(), key=lambda kv: -kv[1][0])[:6]}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
