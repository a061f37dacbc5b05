cd /root/repo
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wn or infer or melenc or flow or rcl or coupling or c2_full" 2>&1 | tail -8 > $O/wn_tests.txt
python tools/step_ab.py > $O/ab_new.json 2> $O/ab.err
SVOC_WN_F25=0 python tools/step_ab.py > $O/ab_old.json 2>> $O/ab.err
python tools/step_ab.py >> $O/ab_new.json 2>> $O/ab.err
SVOC_WN_F25=0 python tools/step_ab.py >> $O/ab_old.json 2>> $O/ab.err
python tools/profile_infer.py 16 512 3 > $O/per_layer_new.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|MOPS" | head -20 > $O/counters.txt
cat $O/wn_tests.txt $O/*.json; grep -E "fusedWN|TOTAL" $O/per_layer_new.txt; cat $O/counters.txt
