cd /root/repo
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1d_winograd or rb1 or resblock" 2>&1 | tail -8 > $O/wino_tests.txt
python tools/wino_bench.py 32 131072 > $O/wino_bench_c32.txt 2>&1
python tools/wino_bench.py 64 65536 > $O/wino_bench_c64.txt 2>&1
for k in 3 7 11; do WL=131072 python tools/wino4_timeline.py 32 $k 1; done > $O/w4_stamps_c32.txt 2>&1
for k in 7 11; do WL=65536 python tools/wino4_timeline.py 64 $k 1; done > $O/w4_stamps_c64.txt 2>&1
python tools/step_ab.py > $O/ab_new.json 2> $O/ab.err
SVOC_W4_C32=0 python tools/step_ab.py > $O/ab_old.json 2>> $O/ab.err
python tools/step_ab.py >> $O/ab_new.json 2>> $O/ab.err
SVOC_W4_C32=0 python tools/step_ab.py >> $O/ab_old.json 2>> $O/ab.err
cat $O/wino_tests.txt $O/wino_bench_c32.txt $O/wino_bench_c64.txt $O/w4_stamps_c32.txt $O/w4_stamps_c64.txt $O/*.json | grep -v amdgpu.ids
