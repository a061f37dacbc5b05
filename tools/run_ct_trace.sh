#!/bin/bash
# kernel durations of the three F(4,2) upsampler shapes of the bench (16 x 512 frames): tail launch on / off, experiment flags
tag=${1:-cttrace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "1 0"; do
  set -- $cfg
  for shape in "512 256 16 8 512" "256 128 16 8 4096"; do
    rm -rf /tmp/kt
    SVOC_CT_TAIL=$1 SVOC_CT_FLAGS=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python $R/tools/convt_probe.py $shape 16 4 > /dev/null 2>&1
    echo "tail=$1 flags=$2 shape=$shape : $(python $R/tools/prof_summary.py /tmp/kt 2>/dev/null | grep 'convt_wino_kernel' | head -1 | awk '{print $(NF-3), $(NF-2), $(NF-1)}')" >> $out/summary.txt
  done
done
cat $out/summary.txt
