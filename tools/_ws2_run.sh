set -x
export SVOC_FUSE=0
for ws in 0 2; do
  for cfg in "128 3 1 32768" "128 7 1 32768" "128 11 1 32768" "128 11 5 32768" "256 3 1 4096" "256 11 5 4096"; do
    SVOC_WS=$ws timeout 120 python tools/conv_probe.py $cfg 16 5 2>&1 | grep -E "^#|conv " | head -4
  done
done
unset SVOC_FUSE
SVOC_WS=2 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "test_infer_vs_reference_golden or test_resblock1 or test_wn or test_generator or test_coupling" 2>&1 | tail -5
