set -x
cd /root/repo
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv1d_winograd" 2>&1 | tail -15 > gpurun_out/r04a/wino_tests.txt
OUT=/tmp/o_new.pt python tools/step_ab.py > gpurun_out/r04a/ab_new.json 2> gpurun_out/r04a/ab_new.err
SVOC_W4_C32=0 OUT=/tmp/o_old.pt python tools/step_ab.py > gpurun_out/r04a/ab_old.json 2>> gpurun_out/r04a/ab_new.err
OUT=/tmp/o_new2.pt python tools/step_ab.py >> gpurun_out/r04a/ab_new.json 2>> gpurun_out/r04a/ab_new.err
SVOC_W4_C32=0 OUT=/tmp/o_old2.pt python tools/step_ab.py >> gpurun_out/r04a/ab_old.json 2>> gpurun_out/r04a/ab_new.err
python - <<'PY' > gpurun_out/r04a/ab_diff.txt 2>&1
import torch
a=torch.load('/tmp/o_new.pt'); b=torch.load('/tmp/o_old.pt')
e=(a-b); print('rel rms new vs old', float(e.pow(2).mean().sqrt()/b.pow(2).mean().sqrt()), 'max', float(e.abs().max()))
PY
cat gpurun_out/r04a/*.json gpurun_out/r04a/ab_diff.txt gpurun_out/r04a/wino_tests.txt
