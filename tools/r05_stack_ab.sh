# round 5: the persistent WN stack launch against one launch per layer (SVOC_WN_STACK=0), one GPU call
cd /root/repo
O=gpurun_out/${1:-r05m}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wn_stack or test_wn or coupling or flow or c2_full_size or full_size_vs_reference" 2>&1 | tail -6 > $O/tests.txt
cat $O/tests.txt
for i in 1 2; do
  python tools/wn_timeline.py 16 512 2>/dev/null | head -2 >> $O/wn_stack.txt
  SVOC_WN_STACK=0 python tools/wn_timeline.py 16 512 2>/dev/null | head -2 >> $O/wn_layers.txt
  python tools/step_ab.py >> $O/step_stack.json 2>> $O/ab.err
  SVOC_WN_STACK=0 python tools/step_ab.py >> $O/step_layers.json 2>> $O/ab.err
done
python tools/profile_infer.py 16 512 3 2>/dev/null | grep -E "WN|TOTAL" > $O/per_layer_stack.txt
echo "== stack"; cat $O/wn_stack.txt $O/step_stack.json $O/per_layer_stack.txt; echo "== per layer"; cat $O/wn_layers.txt $O/step_layers.json
