# A/B of one environment switch on the bench: tools/ab_env.sh VAR [A B]  (runs VAR=A, VAR=B, VAR=A, VAR=B; default 0 1)
cd $GRAFT_REPO_ROOT
A=${2:-0}; B=${3:-1}
for v in $A $B $A $B; do
  echo -n "$1=$v  "
  env $1=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"])"
done
