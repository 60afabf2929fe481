mkdir -p gpurun_out/r6d
MRG_LIB_PATH=$PWD/mirage_amd/libmirage_hip_ts.so python tools/debug/rowfuse_ts.py > gpurun_out/r6d/rowfuse_ts.txt 2>&1
