cd /root/repo
O=gpurun_out/r04_full; mkdir -p $O
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/gpu_tests.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/gpu_tests.txt; cat $O/bench.json | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j['roofline']
print('ms', j['ms_per_step'], 'frac', r['frac'], 'direct', r['frac_direct_form'], 'pmc/lib', r.get('executed_flops_pmc_over_library'), 'traffic', r.get('traffic_bytes_per_step'))
print('dom', r.get('dominant_kernel',{}).get('avg_launch_us'), r.get('dominant_kernel',{}).get('frac'))
print('other', {k:(v['ms_per_step'] if isinstance(v,dict) else v) for k,v in j.get('other_configs',{}).items()})
print('parity', j.get('parity')); print('cpu', j.get('cpu_baseline',{}).get('value'), j.get('cpu_baseline',{}).get('cores'))
"; tail -5 $O/bench.err
