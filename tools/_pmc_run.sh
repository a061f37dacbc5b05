cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SVOC_FUSE=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "MfmaUtil VALUBusy"; do
  d=/tmp/pmc_$(echo $set | tr " " "_" | cut -c1-40)
  rocprofv3 --pmc $set -d $d --output-format csv -- python $R/tools/conv_probe.py 128 11 1 32768 16 3 > /dev/null 2>&1
  echo "== $set"
  python $R/tools/pmc_summary.py $d 2
done
