# A/B of arbitrary env settings: bash tools/debug/run_ab2.sh "c2 c4" reps "ENV1=a ENV2=b" "ENV1=c" ...
cd $GRAFT_REPO_ROOT
CFGS=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do for cfg in $CFGS; do for setting in "$@"; do
  r=$(env $setting timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$cfg [$setting] rep$rep: $r"
done; done; done
