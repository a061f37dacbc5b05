# A/B of the F(4,4) form (k = 7 / 11) against F(4,3): parity slice, step time twice each way, per-launch profile.
# Run on the GPU box: gpurun -- 'bash tools/f44_ab.sh'
cd /root/repo
O=gpurun_out/r04_f44; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv1d_winograd or rb1 or resblock1 or generator or infer_vs_reference or c2_full or c5_full or mrf" 2>&1 | tail -15 > $O/tests.txt
python tools/step_ab.py > $O/ab_f44.json 2> $O/ab.err
SVOC_W4_F44=0 python tools/step_ab.py > $O/ab_f43.json 2>> $O/ab.err
python tools/step_ab.py >> $O/ab_f44.json 2>> $O/ab.err
SVOC_W4_F44=0 python tools/step_ab.py >> $O/ab_f43.json 2>> $O/ab.err
python tools/profile_infer.py 16 512 3 > $O/per_layer_f44.txt 2>&1
SVOC_W4_F44=0 python tools/profile_infer.py 16 512 3 > $O/per_layer_f43.txt 2>&1
cat $O/tests.txt $O/*.json; grep -E "wino4|TOTAL" $O/per_layer_f44.txt; echo; grep -E "wino4|TOTAL" $O/per_layer_f43.txt
