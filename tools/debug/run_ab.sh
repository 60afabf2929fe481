# same-box A/B of one environment knob: bash tools/debug/run_ab.sh KNOB "v1 v2" "c2 c4" [reps]
cd $GRAFT_REPO_ROOT
KNOB=$1; VALS=$2; CFGS=$3; REPS=${4:-2}
for rep in $(seq $REPS); do for cfg in $CFGS; do for v in $VALS; do
  r=$(env $KNOB=$v timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  echo "$cfg $KNOB=$v rep$rep: $r"
done; done; done
