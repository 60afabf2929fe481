python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -5
for r in 1 2 3; do
  echo "new:"; NOBIAS=1 python tools/bench_attn_bwd.py c2 2>/dev/null | grep -v amdgpu
  echo "general kernel:"; MRG_ATTN_GT=0 NOBIAS=1 python tools/bench_attn_bwd.py c2 2>/dev/null | grep -v amdgpu
done
REPS=3 bash tools/ab.sh "c2" "MRG_ATTN_GT=0" "MRG_NOP=1"
