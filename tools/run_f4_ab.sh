# A/B of the F(4,3) Winograd kernels against the F(2,3) ones inside ONE gpurun call:  gpurun -- 'bash tools/run_f4_ab.sh [tag]'
TAG=${1:-r03b}
O=gpurun_out/$TAG
mkdir -p $O
export SVOC_WINO_F4=1   # (the default since r03e)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv1d_winograd or test_resblock1 or test_generator or infer_vs_reference or c2_full_size or c5_full" 2>&1 | tail -15 > $O/f4_tests.txt
cat $O/f4_tests.txt
timeout 300 python tools/wino_bench.py 2>/dev/null > $O/wino_bench_f4.txt
SVOC_WINO_F4=0 timeout 300 python tools/wino_bench.py 2>/dev/null > $O/wino_bench_f2.txt
paste $O/wino_bench_f2.txt $O/wino_bench_f4.txt | awk '{print $1,$4,$5,$6, $(NF/2-1), $(NF/2), "|", $(NF/2+1), $(NF-1), $NF}'
for k in 3 7 11; do timeout 120 python tools/wino4_timeline.py 128 $k 1 2>/dev/null; done | tee $O/wino4_timeline.txt
timeout 120 python tools/wino4_timeline.py 256 11 1 2>/dev/null | tee -a $O/wino4_timeline.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/bench_f4.json 2> $O/bench_f4.err
SVOC_WINO_F4=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/bench_f2.json 2> $O/bench_f2.err
python - <<PY
import json
for t in ("f2","f4"):
    try:
        j=json.load(open("$O/bench_%s.json" % t)); print(t, j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("dominant_kernel",{}).get("avg_launch_us"))
    except Exception as e: print(t, "ERR", e, open("$O/bench_%s.err" % t).read()[-1500:])
PY
