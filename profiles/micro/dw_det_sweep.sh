# step time against the split-K rule of the weight-gradient launch, deterministic seam vs fp32 atomics (gpurun -- bash profiles/micro/dw_det_sweep.sh)
for mt in 16 20 24 28; do for tg in 192 256 384; do
MAGIC_SPLITK_MIN_TILES=$mt MAGIC_SPLITK_TARGET=$tg python bench.py --steps 60 --no-cpu-baseline --no-parity --no-secondary --no-profile > gpurun_out/sw.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/sw.json')); print('det min_tiles $mt target $tg', j['ms_per_step'], j['ms_per_step_steady'])"
done; done
for mt in 12 20 32; do
MAGIC_DW_ATOMICS=1 MAGIC_SPLITK_MIN_TILES=$mt python bench.py --steps 60 --no-cpu-baseline --no-parity --no-secondary --no-profile > gpurun_out/sw.json 2>/dev/null
python -c "
import json; j=json.load(open('gpurun_out/sw.json')); print('atomic min_tiles $mt', j['ms_per_step'], j['ms_per_step_steady'])"
done
