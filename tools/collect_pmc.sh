#!/bin/bash
# PMC passes only (HBM traffic + instruction mix) for the bench workload, single stream.  gpurun -- 'bash tools/collect_pmc.sh <tag>'
TAG=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && python bench.py ) > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > /dev/null 2>&1
python $R/tools/timeline.py /tmp/kt 400 > $O/${TAG}_kernel_timeline_one_step.txt
NK=$(head -1 $O/${TAG}_kernel_timeline_one_step.txt | sed 's/# \([0-9]*\) kernels.*/\1/')
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d /tmp/pr --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d /tmp/pw --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pr /tmp/pw $NK 2 > $O/${TAG}_pmc_hbm_traffic.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d /tmp/pi1 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pi2 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1
python $R/tools/pmc_inst_mix.py /tmp/pi1 /tmp/pi2 > $O/${TAG}_pmc_instruction_mix.txt 2>&1
cat $O/${TAG}_pmc_hbm_traffic.json; cat $O/${TAG}_pmc_instruction_mix.txt; head -c 1500 $O/${TAG}_bench.json
