#!/bin/bash
# round 6: attn_fwd_kernel / attn_fwd_pair_kernel durations (rocprofv3 kernel stats of the bench command) and LDS bank-conflict share (one --pmc pass, eager) with
# the transposed softmax form (this tree) and with the previous library (vln-magic_amd/libmagic_hip_attnold.so, built by hand from the previous attention.hip)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for t in this attnold; do
  if [ $t = this ]; then unset MAGIC_LIB_FILE MAGIC_ALLOW_STALE_LIB; else export MAGIC_LIB_FILE=$R/vln-magic_amd/libmagic_hip_$t.so MAGIC_ALLOW_STALE_LIB=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/abattn_$t -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-secondary --no-profile > /dev/null 2> $O/abattn_$t.err || exit 1
  f=$(find $O/abattn_$t -name "*kernel_stats.csv" | head -1)
  echo "== $t: kernel stats" | tee -a $O/r06_ab_attn_fwd.txt
  grep "attn_fwd" $f | awk -F, '{printf "%s calls %s avg %.1f us\n", substr($1,1,48), $2, $4/1000}' | tee -a $O/r06_ab_attn_fwd.txt
  rm -rf $O/abattn_$t
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/abattn_pmc_$t -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-parity --no-secondary > /dev/null 2> $O/abattn_pmc_$t.err || exit 1
  python3 - $O/abattn_pmc_$t <<'PY' | tee -a $O/r06_ab_attn_fwd.txt
import csv, glob, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "attn_" not in k: continue
        k = k.split("(")[0][:40]
        a = acc.setdefault(k, {})
        a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, a in sorted(acc.items()):
    print(f"   {k:40s} lds_bank_conflict / lds_idx_active = {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
  rm -rf $O/abattn_pmc_$t
done
