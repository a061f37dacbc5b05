# A/B of two builds of the library inside ONE gpurun call: the working tree's libsvoc_hip.so against csrc/libsvoc_hip_ab.so (built from
# another commit with `git archive <rev> smart-vocoder_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/smart-vocoder_amd/csrc`).
#   gpurun -- 'bash tools/lib_ab.sh [tag] [profile-filter-regex]'
cd /root/repo
TAG=${1:-lib_ab}; FLT=${2:-"wino4|TOTAL"}
OLD=/root/repo/smart-vocoder_amd/csrc/libsvoc_hip_ab.so
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "conv1d_winograd or rb1 or resblock1 or generator or infer_vs_reference or c2_full or mrf" 2>&1 | tail -5 > $O/tests.txt
for i in 1 2; do
  python tools/step_ab.py >> $O/new.json 2>> $O/ab.err
  SVOC_LIB=$OLD python tools/step_ab.py >> $O/old.json 2>> $O/ab.err
done
python tools/profile_infer.py 16 512 3 > $O/per_layer_new.txt 2>&1
SVOC_LIB=$OLD python tools/profile_infer.py 16 512 3 > $O/per_layer_old.txt 2>&1
cat $O/tests.txt
echo "== new"; cat $O/new.json; grep -E "$FLT" $O/per_layer_new.txt
echo "== old"; cat $O/old.json; grep -E "$FLT" $O/per_layer_old.txt
