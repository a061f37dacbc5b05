cd /root/repo
O=gpurun_out/r04q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rb1 or generator or infer_vs_reference or c2_full" 2>&1 | tail -4 > $O/tests.txt
python tools/step_ab.py > $O/ab_new.json 2> $O/ab.err
SVOC_W4_PAIR=0 python tools/step_ab.py > $O/ab_old.json 2>> $O/ab.err
python tools/step_ab.py >> $O/ab_new.json 2>> $O/ab.err
SVOC_W4_PAIR=0 python tools/step_ab.py >> $O/ab_old.json 2>> $O/ab.err
python tools/profile_infer.py 16 512 3 > $O/per_layer_new.txt 2>&1
SVOC_W4_PAIR=0 python tools/profile_infer.py 16 512 3 > $O/per_layer_old.txt 2>&1
cat $O/tests.txt $O/*.json; grep -E "wino4|TOTAL" $O/per_layer_new.txt; echo; grep -E "wino4|TOTAL" $O/per_layer_old.txt
