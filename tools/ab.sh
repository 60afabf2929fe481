#!/bin/bash
# same-box A/B of bench.py under environment settings: tools/ab.sh "<config> [config ...]" "ENV=a ENV2=b" "ENV=c" ...  (each quoted group is one arm)
# an arm that starts with BASE runs the baseline worktree .ab_base (git worktree add .ab_base <rev>; build its csrc; keep it out of commits
# with `echo .ab_base/ >> .git/info/exclude`, remove it with `git worktree remove --force .ab_base`) with the rest as env
# arms run interleaved, REPS (default 2) repetitions; prints ms_per_step per arm and repetition
cfgs="$1"; shift
reps=${REPS:-2}
steps=${STEPS:-30}
for cfg in $cfgs; do
  for r in $(seq 1 $reps); do
    i=0
    for arm in "$@"; do
      i=$((i+1))
      dir=.
      case "$arm" in BASE*) dir=.ab_base; arm="${arm#BASE}"; arm="${arm:-MRG_NOP=1}";; esac
      ms=$(cd $dir && env $arm python bench.py --config $cfg --steps $steps --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")
      echo "$cfg rep$r arm$i [$arm] $ms ms"
    done
  done
done
