"""MFMA utilisation of the dense-contraction kernels from a rocprofv3 PMC pass (north star: 'rocprof ... MFMA utilisation against chip peak').

On the GPU box:
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile
  python3 $R/profiles/pmc_mfma.py $R/gpurun_out/pmc_mfma > $R/gpurun_out/pmc_mfma.json
Utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 256 CUs x 4 SIMDs) (the gfx94x `MfmaUtil` formula; ROCm 7.2 ships no gfx950
derived-counter section, MI355X_MICROARCH.md 'rocprofv3 PMC slots').  GRBM_GUI_ACTIVE is reported SUMMED over the 8 XCDs (it reads
~370 k cycles for a ~19 us dispatch), so active cycles = GRBM_GUI_ACTIVE / 8.  Cross-check printed with it: busy cycles / 16 = number of
v_mfma_f32_16x16x32_bf16 instructions (16 issue cycles each, MI355X_MICROARCH.md 'Per-instruction cycle constants'), x 16 384 FLOP each
must reproduce the algorithmic FLOPs bench.py counts (145.7 GFLOP per step over 155 eager GEMM launches = 0.94 GFLOP per launch)."""
import csv
import glob
import json
import os
import sys

CUS = 256


def main():
    d = sys.argv[1]
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                key = (row["Dispatch_Id"], row["Kernel_Name"])
                per.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
    fam = {}
    for (_, name), c in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        k = ("gemm (gemm_kernel / gemm_xcd_kernel / gemm_grouped_kernel)" if ("gemm_kernel" in name or "gemm_xcd" in name or "gemm_grouped" in name)
             else "linear_ln / linear_lnbwd" if "linear_ln" in name else "attention fwd/bwd" if "attn_" in name else None)
        if k is None:
            continue
        a = fam.setdefault(k, [0.0, 0.0, 0])
        a[0] += c["SQ_VALU_MFMA_BUSY_CYCLES"]
        a[1] += c["GRBM_GUI_ACTIVE"]
        a[2] += 1
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, bench.py --mode eager (bf16, B=48)",
           "formula": "SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 256 CUs * 4 SIMDs); counters collected with dispatches serialised",
           "families": {k: {"dispatches": n, "mfma_busy_cycles_per_dispatch": round(b / n), "active_cycles_per_dispatch": round(g / 8 / n),
                            "mfma_util": round(b / (g / 8 * CUS * 4), 5) if g else None,
                            "implied_gflop_per_dispatch": round(b / n / 16 * 16384 / 1e9, 3)}
                        for k, (b, g, n) in fam.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
