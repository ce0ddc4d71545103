"""Per-kernel counter summary for the kernels that carry the step (round 4; north star: "rocprof HBM GB/s and MFMA utilisation against chip
peak"): merges several `rocprofv3 --pmc` passes of the SAME command with the `--kernel-trace --stats` CSV of that command into one row per
kernel name.

  python3 profiles/pmc_kernels.py <kernel_stats.csv> <pmc_dir> [<pmc_dir> ...] > profiles/r04_pmc_kernels.json

Passes (each its own run: `rocprofv3 --pmc ... --output-format csv -d <dir> -- python3 bench.py --mode eager --steps 6 --warmup 3 ...`;
the program sits directly after `--`; --pmc is never combined with a trace option):
  A  SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE                                  -> MFMA utilisation
  B  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY   -> LDS conflicts, where wave cycles go
  C  FETCH_SIZE      D  WRITE_SIZE   (TCC: 3 + 2 slots of 4, separate passes)   -> HBM-side bytes
Formulas and corrections (MI355X_MICROARCH.md): mfma_util = MFMA busy cycles / ((GRBM_GUI_ACTIVE / 8 XCDs) x 256 CUs x 4 SIMDs) -- counters
are collected with dispatches serialised, so this is the kernel ALONE on the chip; a v_mfma_f32_16x16x32_{bf16,f16} holds the pipe 16 cycles
(8 passes x ... ) and does 16 384 FLOP, so implied GFLOP = busy / 16 x 16 384 / 1e9; FETCH_SIZE / WRITE_SIZE are in KB and FETCH_SIZE is
doubled on gfx950 (it tallies 128-byte requests at 64 bytes); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles /
all LDS-array cycles); wait / issue-stall / active fractions are of SQ_WAVE_CYCLES (disjoint buckets)."""
import csv
import glob
import json
import os
import re
import sys

CUS, PEAK_TFLOPS, PEAK_GBS = 256, 2500.0, 8000.0
KEEP = ("chain64_fwd_kernel", "chain_fwd_kernel", "encoder_mix_kernel", "encoder_rs_kernel", "encoder_fwd_kernel", "xencoder_rs_kernel", "xencoder_fwd_kernel",
        "rowbwd16a_kernel", "rowbwd16_kernel", "rowbwd32_kernel", "rowbwd64_kernel", "attn_bwd", "attn_fwd", "gemm_dw_batch_kernel", "gemm_grouped_kernel",
        "gemm_xcd_kernel", "gemm_kernel", "linear_ln", "mlm_", "adamw_kernel", "mse_multi", "ln_bwd", "ln_fwd", "view_gather",
        # the navigator iteration at MAGIC-L width (final_profile_r05.sh navpmc)
        "gemm_kg_kernel", "gemm_wide_kernel", "gemm_dw_cat_kernel", "gemm_grouped_kg_kernel", "smallk_ln", "pano_fuse", "colsum")


def short(name):
    """kernel name without template arguments / parameter lists (rocprofv3 prints some names mangled: _Z<len><name>...)"""
    m = re.match(r"_Z(?:N\w*?)?(\d+)", name)
    if m:
        i = m.end()
        return name[i:i + int(m.group(1))]
    n = re.sub(r"\(.*$", "", name)
    n = re.sub(r"<.*$", "", n)
    return n.replace("void ", "").strip()


def load_pmc(dirs):
    per = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    c = per.setdefault(k, {})
                    a = c.setdefault(row["Counter_Name"], [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    return per


def load_stats(path):
    out = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Name"])
            a = out.setdefault(k, [0.0, 0])
            a[0] += float(row["TotalDurationNs"])
            a[1] += int(row["Calls"])
    return out


def main():
    stats = load_stats(sys.argv[1])
    per = load_pmc(sys.argv[2:])
    rows = {}
    for k, c in per.items():
        if not any(x in k for x in KEEP):
            continue
        avg = lambda name: (c[name][0] / c[name][1]) if name in c and c[name][1] else None
        r = {"dispatches_counted": max(v[1] for v in c.values())}
        if k in stats:
            r["avg_us_kernel_trace"] = round(stats[k][0] / stats[k][1] / 1e3, 2)
            r["calls_kernel_trace"] = stats[k][1]
        busy, act = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("GRBM_GUI_ACTIVE")
        if busy is not None and act:
            cyc = act / 8.0
            r["mfma_busy_cycles"] = round(busy)
            r["active_cycles"] = round(cyc)
            r["mfma_util"] = round(busy / (cyc * CUS * 4), 5)
            r["implied_gflop_per_dispatch"] = round(busy / 16 * 16384 / 1e9, 3)
            if "avg_us_kernel_trace" in r:
                t = r["implied_gflop_per_dispatch"] / (r["avg_us_kernel_trace"] * 1e-6) / 1e3
                r["issued_tflops_on_trace_time"] = round(t, 1)
                r["frac_of_bf16_peak"] = round(t / PEAK_TFLOPS, 4)
        conf, ldsa = avg("SQ_LDS_BANK_CONFLICT"), avg("SQ_LDS_IDX_ACTIVE")
        if conf is not None and ldsa:
            r["lds_bank_conflict_frac"] = round(conf / ldsa, 4)
        wc = avg("SQ_WAVE_CYCLES")
        if wc:
            for name, key in (("SQ_WAIT_ANY", "wave_cycles_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stall_frac"),
                              ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing_frac")):
                v = avg(name)
                if v is not None:
                    r[key] = round(v / wc, 4)
        fe, wr = avg("FETCH_SIZE"), avg("WRITE_SIZE")
        if fe is not None:
            r["hbm_fetch_bytes"] = round(fe * 1024 * 2)
        if wr is not None:
            r["hbm_write_bytes"] = round(wr * 1024)
        if fe is not None and wr is not None and "avg_us_kernel_trace" in r:
            gbs = (r["hbm_fetch_bytes"] + r["hbm_write_bytes"]) / (r["avg_us_kernel_trace"] * 1e-6) / 1e9
            r["hbm_GBps_on_trace_time"] = round(gbs, 1)
            r["frac_of_hbm_peak"] = round(gbs / PEAK_GBS, 4)
        rows[k] = r
    order = sorted(rows, key=lambda k: -(rows[k].get("avg_us_kernel_trace", 0) * rows[k].get("calls_kernel_trace", 0)))
    print(json.dumps({"source": "rocprofv3 --pmc passes A-D (bench.py --mode eager, dispatches serialised by the collection) merged with the "
                                "--kernel-trace --stats CSV of the graph-replay run; see the header of profiles/pmc_kernels.py for formulas",
                      "peaks": {"bf16_mfma_tflops": PEAK_TFLOPS, "hbm_GBps": PEAK_GBS}, "kernels": {k: rows[k] for k in order}}, indent=1))


if __name__ == "__main__":
    main()
