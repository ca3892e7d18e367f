# A/B of one environment switch on the bench: tools/ab_env.sh VAR  (runs VAR=0, VAR=1, VAR=0, VAR=1)
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  env $1=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --sample-steps 0 2>&1 | tail -1 | cut -c1-130
done
