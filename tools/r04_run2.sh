cd /root/repo
O=gpurun_out/r04b; mkdir -p $O
python tools/wino_bench.py 32 131072 > $O/wino_bench_c32.txt 2>&1
for k in 3 7 11; do WL=131072 python tools/wino4_timeline.py 32 $k 1; done > $O/w4_stamps_c32.txt 2>&1
WL=131072 python tools/wino4_timeline.py 32 11 3 >> $O/w4_stamps_c32.txt 2>&1
python tools/profile_infer.py 16 512 3 > $O/per_layer_new.txt 2>&1
SVOC_W4_C32=0 python tools/profile_infer.py 16 512 3 > $O/per_layer_old.txt 2>&1
cat $O/wino_bench_c32.txt $O/w4_stamps_c32.txt; grep -E "Ci32 |C32|rb32|resblock|fused" $O/per_layer_new.txt | head -30; grep -E "Ci32|C32|fused|resblock" $O/per_layer_old.txt | head
