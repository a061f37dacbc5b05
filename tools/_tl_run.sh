cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/tl --output-format csv -- python $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/timeline.py /tmp/tl 400 > $R/gpurun_out/timeline.txt
head -5 $R/gpurun_out/timeline.txt
