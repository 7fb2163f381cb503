#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r2z; mkdir -p $O
cd /tmp
for gb in 32 64; do
VLB_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace -d $O/prof$gb -o r -- python $R/bench.py --global-batch $gb --steps 10 --warmup 2 --no-cpu-baseline --no-phase-times --head-start 0 > $O/prof$gb.log 2>&1
echo "== batch $gb"; python $R/tools/kstats.py $O/prof$gb 13 30 | tee $O/kstats_$gb.txt
rm -rf $O/prof$gb
done
