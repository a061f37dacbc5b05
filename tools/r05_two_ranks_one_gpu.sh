# two ranks of bench.py on ONE GPU (gloo): does parallel.infer_sharded return the timed step's bits?  (persistent WN stack launch under GPU sharing)
cd /root/repo
for v in "SVOC_WN_STACK=1" "SVOC_WN_STACK=0"; do
  for rep in 1 2; do
  echo "== $v"
  env $v BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['ms_per_step'], j['per_rank']['infer_ms'], j['infer_sharded'].get('equals_timed_step_output'), j['infer_sharded'].get('ms_per_call'))
"
  done
done
