mkdir -p gpurun_out/knobs
REPS=2 bash tools/ab.sh "c2 c4" "MRG_SIDE_BLOCKS=6" "MRG_SIDE_BLOCKS=4" "MRG_SIDE_BLOCKS=8" "MRG_SIDE_BLOCKS=12" "MRG_SIDE_BLOCKS=3" > gpurun_out/knobs/side_blocks.txt 2>&1
