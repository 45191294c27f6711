# A/B of one environment variable's values on ONE box: bash scripts/ab_envval.sh VAR v1 v2 ...   (bench.py, two rounds)
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.log
  echo "$VAR=$v: $(grep -o 'timed region done: [0-9.]* ms/step' /tmp/b.log) $(grep -o 'stash mode: [0-9.]* ms/step' /tmp/b.log)"
done; done
