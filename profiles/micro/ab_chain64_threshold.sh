cd $GRAFT_REPO_ROOT
F="--steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-secondary --no-profile"
show='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["steady"]["ms_per_step"])'
for i in 1 2 3; do
  for v in 0 100 257 100000; do
    MAGIC_CHAIN_64_MIN_TILES=$v python bench.py $F 2>/dev/null | python -c "$show" "min_tiles=$v"
  done
done
