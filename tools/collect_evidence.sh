#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/evidence/ (copy what should be judged to profiles/).
#   gpurun --timeout 2400 -- 'bash tools/collect_evidence.sh <tag>'
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) > $O/${TAG}_gpu_tests.txt
( cd $R && python bench.py ) > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
STEPS=5; WARM=2
rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-pmc --no-other-configs > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/kt > $O/${TAG}_bench_kernel_trace_summary.txt
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/${TAG}_rocprofv3_kernel_stats.csv
python $R/tools/timeline.py /tmp/kt 400 > $O/${TAG}_kernel_timeline_one_step.txt
NK=$(head -1 $O/${TAG}_kernel_timeline_one_step.txt | sed 's/# \([0-9]*\) kernels.*/\1/')
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/pr --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-other-configs > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d /tmp/pw --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-other-configs > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pr /tmp/pw $NK 2 > $O/${TAG}_pmc_hbm_traffic.json
( cd $R && python tools/profile_infer.py 16 512 3 ) > $O/${TAG}_per_layer_event_profile.txt 2>&1
( cd $R && python tools/latency_probe.py ) > $O/${TAG}_latency_by_shape.txt 2>&1


( cd $R && for k in 3 7 11; do python tools/wino4_timeline.py 128 $k 1; done; python tools/wino4_timeline.py 128 11 3; python tools/wino4_timeline.py 256 11 1; for k in 7 11; do WL=65536 python tools/wino4_timeline.py 64 $k 1; done; for k in 3 7 11; do WL=131072 python tools/wino4_timeline.py 32 $k 1; done; WL=131072 python tools/wino4_timeline.py 32 11 3 ) > $O/${TAG}_winograd_workgroup_stamps.txt 2>/dev/null
( cd $R && for sh in "512 256 16 8 512" "256 128 16 8 4096" "128 64 4 2 32768" "64 32 4 2 65536"; do python tools/ct_timeline.py $sh; done;  ) > $O/${TAG}_upsampler_f42_workgroup_stamps.txt 2>/dev/null
( cd $R && for v in 1 0; do SVOC_CT_WINO=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | tail -1; done ) > $O/${TAG}_bench_upsamplers_f42_vs_direct.txt
( cd $R && python tools/wn_timeline.py 16 512 | head -1; echo "(above: the persistent stack launch, csrc/wn_stack.hip, default; below: one launch per layer, SVOC_WN_STACK=0, whose kernel carries the phase stamps)"; SVOC_WN_STACK=0 python tools/wn_timeline.py 16 512 ) > $O/${TAG}_wn_layer_phase_stamps.txt 2>/dev/null
( cd $R && python tools/wino_bench.py 128 32768; python tools/wino_bench.py 64 65536; python tools/wino_bench.py 256 4096; python tools/wino_bench.py 32 131072 ) > $O/${TAG}_winograd_per_conv.txt 2>/dev/null
( cd $R && BENCH_BACKEND=nccl BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29573 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-other-configs ) > $O/${TAG}_bench_1rank_rccl.json 2> $O/${TAG}_bench_1rank_rccl.err
# instruction mix of the DEFAULT plan (separate counter passes, nothing traced beside them)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d /tmp/pi1 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-other-configs > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pi2 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-other-configs > /dev/null 2>&1
python $R/tools/pmc_inst_mix.py /tmp/pi1 /tmp/pi2 > $O/${TAG}_pmc_instruction_mix.txt 2>&1
ls -la $O

( cd $R && python tools/profile_infer.py 1 200 5 ) > $O/${TAG}_per_layer_event_profile_1x200.txt 2>&1
( cd $R && BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 5 --warmup 2 ) > $O/${TAG}_bench_2ranks_gloo_one_gpu.json 2> $O/${TAG}_bench_2ranks_gloo_one_gpu.err
ls -la $O
# round 5: clock / power study (steady state: every workload looped back to back for seconds, board power from rocm-smi)
( cd $R && timeout 300 tools/clock_power_probe ) > $O/${TAG}_clock_power_probe.txt 2>&1
( cd $R && timeout 400 python tools/power_ablate.py ) > $O/${TAG}_power_ablation_steady_state.txt 2>&1
( cd $R && timeout 300 python tools/power_watch.py ) > $O/${TAG}_power_watch.txt 2>&1
( cd $R && timeout 300 python tools/power_wn.py ) > $O/${TAG}_power_wn_stack.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt200 --output-format csv -- python $R/tools/small_shape_run.py 1 200 20 > /dev/null 2>&1; python $R/tools/timeline.py /tmp/kt200 400 > $O/${TAG}_kernel_timeline_1x200_graph.txt 2>&1 )
ls -la $O
# round 5: the short-input persistent WN launch (phase stamps) and both persistent launches with several processes on the one GPU
( cd $R && python tools/wn_mesh_timeline.py 1 200 16; python tools/wn_mesh_timeline.py 1 200 8; python tools/wn_mesh_timeline.py 2 150 8; python tools/wn_mesh_timeline.py 1 512 16 ) > $O/${TAG}_wn_mesh_phase_stamps.txt 2>/dev/null
( cd $R && for n in 2 4; do echo "== $n processes, 1 x 200 (csrc/wn_mesh.hip)"; timeout 200 python tools/wn_stack_shared_gpu.py $n 2000 1 200; done; echo "== 2 processes, 1 x 512 (csrc/wn_mesh.hip, two tiles per group)"; timeout 200 python tools/wn_stack_shared_gpu.py 2 2000 1 512; for n in 2 4; do echo "== $n processes, 16 x 512 (csrc/wn_stack.hip)"; timeout 300 python tools/wn_stack_shared_gpu.py $n 200; done ) > $O/${TAG}_persistent_launches_shared_gpu.txt 2>&1
ls -la $O
# round 6: the WN stack launch looped alone / with its weights evicted (what the prefetch pass at the head of infer buys), mid-size batches, the device-side give-up tests
( cd $R && python tools/wn_in_step_probe.py ) > $O/${TAG}_wn_in_step_probe.txt 2>&1
( cd $R && python tools/latency_probe.py 2x512 3x512 4x512 5x512 6x512 8x512 ) > $O/${TAG}_latency_mid_size.txt 2>&1
( cd $R && python -m pytest tests/test_gpu_parity.py -q -m gpu -k "give_up or persistent" 2>&1 | tail -3 ) > $O/${TAG}_give_up_tests.txt
ls -la $O
# round 6: one replayed 4 x 512 step kernel by kernel (the mid-size regime: short-input WN chain, the dilated rows' tail launches)
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt4 --output-format csv -- python $R/tools/small_shape_run.py 4 512 20 > /dev/null 2>&1; python $R/tools/timeline.py /tmp/kt4 400 > $O/${TAG}_kernel_timeline_4x512.txt 2>&1 )
ls -la $O
