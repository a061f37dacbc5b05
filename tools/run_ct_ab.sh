#!/bin/bash
# F(4,2) upsampler kernel against the direct polyphase kernel inside one GPU call: parity slice, bench both ways.
tag=${1:-ct}
out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_transpose or test_generator or infer_vs or c2_full or variant_batch or shard" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
for v in 1 0; do
  for i in 1 2; do
    SVOC_CT_WINO=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-pmc 2>/dev/null | tail -1 > $out/bench_ct${v}_$i.json
    python - <<PY
import json; d=json.load(open("$out/bench_ct${v}_$i.json")); print("CT_WINO=$v", d["ms_per_step"], d.get("parity"))
PY
  done
done
SVOC_CT_WINO=1 timeout 300 python tools/profile_infer.py 16 512 3 > $out/profile_ct1.txt 2>&1
grep -i "convT" $out/profile_ct1.txt
SVOC_CT_WINO=0 timeout 300 python tools/profile_infer.py 16 512 3 > $out/profile_ct0.txt 2>&1
grep -i "convT" $out/profile_ct0.txt
