R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d /tmp/pi1 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pi2 --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1
python $R/tools/pmc_inst_mix.py /tmp/pi1 /tmp/pi2 > $O/r03_n_pmc_instruction_mix.txt 2>&1
cat $O/r03_n_pmc_instruction_mix.txt
