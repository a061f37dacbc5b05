# F(4,4) on the 64- / 32-row layouts (single-convolution kernels, dilation 1; SVOC_W4_F44=2) against F(4,3): per-launch time and stamps.
cd /root/repo
O=gpurun_out/r04_f44s; mkdir -p $O
export WB_D=1 WB_K=7,11
for F in 2 1; do
  SVOC_W4_F44=$F timeout 200 python tools/wino_bench.py 64 65536 16 2>/dev/null > $O/c64_f$F.txt
  SVOC_W4_F44=$F timeout 200 python tools/wino_bench.py 32 131072 16 2>/dev/null > $O/c32_f$F.txt
  for k in 7 11; do SVOC_W4_F44=$F WL=65536 timeout 120 python tools/wino4_timeline.py 64 $k 1 2>/dev/null; done > $O/stamps_c64_f$F.txt
  for k in 7 11; do SVOC_W4_F44=$F WL=131072 timeout 120 python tools/wino4_timeline.py 32 $k 1 2>/dev/null; done > $O/stamps_c32_f$F.txt
done
echo "C=64 F(4,4) | F(4,3)"; paste $O/c64_f2.txt $O/c64_f1.txt; echo "C=32"; paste $O/c32_f2.txt $O/c32_f1.txt
echo "stamps F(4,4)"; cat $O/stamps_c64_f2.txt $O/stamps_c32_f2.txt; echo "stamps F(4,3)"; cat $O/stamps_c64_f1.txt $O/stamps_c32_f1.txt
SVOC_W4_F44=2 timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv1d_winograd and (64-64 or 32-32)" 2>&1 | tail -5
