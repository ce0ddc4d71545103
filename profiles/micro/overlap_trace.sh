#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ovt
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ovt -- python3 $R/profiles/micro/overlap_trace.py > $R/gpurun_out/overlap_trace.txt 2> $R/gpurun_out/overlap_trace.err
f=$(find $R/gpurun_out/ovt -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/micro/overlap_trace_report.py "$f" >> $R/gpurun_out/overlap_trace.txt 2>&1
rm -rf $R/gpurun_out/ovt
cat $R/gpurun_out/overlap_trace.txt
