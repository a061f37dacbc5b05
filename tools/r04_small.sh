cd /root/repo
O=gpurun_out/r04m; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wn or infer or melenc or flow or rcl or coupling or small_shape or posterior or variant_batch or plan" 2>&1 | tail -8 > $O/tests.txt
python tools/latency_probe.py 2>&1 | grep -v amdgpu > $O/lat_new.txt
SVOC_WN_SMALL_F25=0 python tools/latency_probe.py 2>&1 | grep -v amdgpu > $O/lat_old.txt
python tools/profile_infer.py 1 200 5 > $O/per_layer_1x200.txt 2>&1
cat $O/tests.txt $O/lat_new.txt $O/lat_old.txt; grep -E "smallWN|conv  Ci192" $O/per_layer_1x200.txt
