# A/B of the F(4,4) form (128-row layout, k = 7 / 11) against F(4,3): parity slice, step time twice each way, per-launch profile.
# Run on the GPU box: gpurun -- 'bash tools/f44_ab.sh'
cd /root/repo
O=gpurun_out/r04_f44; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1d_winograd or rb1 or resblock1 or generator or infer_vs_reference or c2_full" 2>&1 | tail -15 > $O/tests.txt
python tools/step_ab.py > $O/ab_f44.json 2> $O/ab.err
SVOC_W4_F44=0 python tools/step_ab.py > $O/ab_f43.json 2>> $O/ab.err
python tools/step_ab.py >> $O/ab_f44.json 2>> $O/ab.err
SVOC_W4_F44=0 python tools/step_ab.py >> $O/ab_f43.json 2>> $O/ab.err
python tools/profile_infer.py 16 512 3 > $O/per_layer_f44.txt 2>&1
SVOC_W4_F44=0 python tools/profile_infer.py 16 512 3 > $O/per_layer_f43.txt 2>&1
cat $O/tests.txt $O/*.json; grep -E "wino4|TOTAL" $O/per_layer_f44.txt; echo; grep -E "wino4|TOTAL" $O/per_layer_f43.txt
# per convolution at the C = 128 / 256 stage sizes, and the stamped build's phase split of one k = 11 / k = 7 tile
timeout 300 python tools/wino_bench.py 128 32768 16 2>/dev/null > $O/wino_bench_c128_f44.txt
SVOC_W4_F44=0 timeout 300 python tools/wino_bench.py 128 32768 16 2>/dev/null > $O/wino_bench_c128_f43.txt
timeout 300 python tools/wino_bench.py 256 4096 16 2>/dev/null > $O/wino_bench_c256_f44.txt
SVOC_W4_F44=0 timeout 300 python tools/wino_bench.py 256 4096 16 2>/dev/null > $O/wino_bench_c256_f43.txt
echo "C=128 F(4,4) | F(4,3)"; paste $O/wino_bench_c128_f44.txt $O/wino_bench_c128_f43.txt
echo "C=256 F(4,4) | F(4,3)"; paste $O/wino_bench_c256_f44.txt $O/wino_bench_c256_f43.txt
for k in 7 11; do timeout 120 python tools/wino4_timeline.py 128 $k 1 2>/dev/null; done | tee $O/wino4_timeline_f44.txt
for k in 7 11; do SVOC_W4_F44=0 timeout 120 python tools/wino4_timeline.py 128 $k 1 2>/dev/null; done | tee $O/wino4_timeline_f43.txt
