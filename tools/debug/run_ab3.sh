# same-box A/B of several knobs: bash tools/debug/run_ab3.sh "c2 c4" reps  KNOB=v ... (each argument after the first two is one setting; "-" = defaults)
cd $GRAFT_REPO_ROOT
CFGS=$1; REPS=$2; shift 2
for rep in $(seq $REPS); do for cfg in $CFGS; do for setting in "$@"; do
  if [ "$setting" = "-" ]; then envs=""; else envs=$(echo $setting | tr ',' ' '); fi
  r=$(env $envs timeout 300 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$cfg [$setting] rep$rep: $r"
done; done; done
