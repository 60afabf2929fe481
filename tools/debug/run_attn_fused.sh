cd $GRAFT_REPO_ROOT
timeout 120 python tools/debug/attn_ts.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_attn_ts.log
