cd $GRAFT_REPO_ROOT
cp mirage_amd/libmirage_hip.so /tmp/lib_keep.so
touch mirage_amd/csrc/attention.hip
make -C mirage_amd/csrc EXTRA="-DMRG_ATTN_TS" > /dev/null 2>&1
timeout 120 python tools/debug/attn_ts.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_ts_new.log
cp /tmp/lib_keep.so mirage_amd/libmirage_hip.so
