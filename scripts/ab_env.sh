# A/B of one environment switch on ONE box: bash scripts/ab_env.sh VAR   (bench.py with VAR unset / VAR=1, twice)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" 1; do
  if [ -z "$v" ]; then unset $1; else export $1=$v; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.log
  echo "$1=${v:-unset}: $(grep -o 'timed region done: [0-9.]* ms/step' /tmp/b.log) $(grep -o 'stash mode: [0-9.]* ms/step' /tmp/b.log)"
done; done
