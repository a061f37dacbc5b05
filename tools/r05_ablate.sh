# round 5: what pulls the clock down inside the Winograd kernel?  Stamped C=128 k=11 d=1 kernel with work removed (SVOC_DBG_ABL bits:
# 1 no global loads, 2 no publish/transform, 4 no epilogue, 8 weights not reloaded, 16 B fragments not re-read); also B=2 (tensors in the MALL)
cd /root/repo
O=gpurun_out/${1:-r05g}; mkdir -p $O
for a in 0 1 2 4 8 16 3 7 15 31 24; do
  echo "== SVOC_DBG_ABL=$a" >> $O/ablate.txt
  SVOC_DBG_ABL=$a python tools/wino4_timeline.py 128 11 1 2>/dev/null | tail -4 >> $O/ablate.txt
done
echo "== B=2 (tensors fit the Infinity Cache)" >> $O/ablate.txt
WB=2 python tools/wino4_timeline.py 128 11 1 2>/dev/null | tail -4 >> $O/ablate.txt
echo "== k=3 abl 0 / 31" >> $O/ablate.txt
SVOC_DBG_ABL=0 python tools/wino4_timeline.py 128 3 1 2>/dev/null | tail -4 >> $O/ablate.txt
SVOC_DBG_ABL=31 python tools/wino4_timeline.py 128 3 1 2>/dev/null | tail -4 >> $O/ablate.txt
cat $O/ablate.txt
