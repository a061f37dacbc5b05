TAG=${1:-r03f}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv1d_winograd or test_resblock1 or test_generator or infer_vs_reference or c2_full_size or c5_full" 2>&1 | tail -15 > $O/f4_tests.txt
cat $O/f4_tests.txt
for C in 64 128; do
L=$((C==64 ? 65536 : 32768))
timeout 300 python tools/wino_bench.py $C $L 2>/dev/null > $O/wino_bench_f4_c$C.txt
SVOC_WINO_F4=0 timeout 300 python tools/wino_bench.py $C $L 2>/dev/null > $O/wino_bench_f2_c$C.txt
paste $O/wino_bench_f2_c$C.txt $O/wino_bench_f4_c$C.txt | awk '{print $1,$2,$4,$5,$6, $(NF/2-1), $(NF/2), "|", $(NF/2+1), $(NF-1), $NF}'
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f4.json 2> $O/bench_f4.err
SVOC_WINO_F4=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f2.json 2> $O/bench_f2.err
python - <<PY
import json
for t in ("f2","f4"):
    try:
        j=json.load(open("$O/bench_%s.json" % t)); print(t, j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("dominant_kernel",{}).get("avg_launch_us"))
    except Exception as e: print(t, "ERR", e, open("$O/bench_%s.err" % t).read()[-1500:])
PY
python tools/profile_infer.py 16 512 3 > $O/per_layer.txt 2>&1; head -60 $O/per_layer.txt
