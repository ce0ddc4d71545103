#!/bin/bash
# same-box comparison of several builds of the library (untracked variant files vln-magic_amd/libmagic_hip_<tag>.so built by hand from
# variant sources): bash profiles/micro/ab_libs.sh rounds tag [tag ...]     ("this" = the tree's own build)
cd $GRAFT_REPO_ROOT
F="--steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-secondary --no-profile"
show='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["steady"]["ms_per_step"])'
n=$1; shift
for i in $(seq 1 $n); do
  for t in "$@"; do
    if [ $t = this ]; then python bench.py $F 2>/dev/null | python -c "$show" this
    else MAGIC_LIB_FILE=$GRAFT_REPO_ROOT/vln-magic_amd/libmagic_hip_$t.so MAGIC_ALLOW_STALE_LIB=1 python bench.py $F 2>/dev/null | python -c "$show" $t; fi
  done
done
