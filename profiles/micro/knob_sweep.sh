# one bench line per knob setting (headline step, graph mode): ms_per_step
mkdir -p gpurun_out
run() { r=$(env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 60 --warmup 12 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); echo "$* ms_per_step=$r" | tee -a gpurun_out/knob_sweep.log; }
run X=0
run MAGIC_LLN_MAXK=1024
run MAGIC_LLN_MAXK=256
run MAGIC_FORCE_SPLIT_GRAPH=1
run MAGIC_DW_SIDE=1
run MAGIC_GEMM_XCD=0
run MAGIC_NO_FUSED_LNB=1
