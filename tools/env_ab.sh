# Generic A/B of one environment switch inside ONE gpurun call: step time twice each way and the per-launch profile of both arms.
#   gpurun -- 'bash tools/env_ab.sh SVOC_W4_ACC3_PIPE=1 [tag] [profile-filter-regex]'
cd /root/repo
SW=$1; TAG=${2:-ab}; FLT=${3:-"wino4|TOTAL"}
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2; do
  env $SW python tools/step_ab.py >> $O/on.json 2>> $O/ab.err
  python tools/step_ab.py >> $O/off.json 2>> $O/ab.err
done
env $SW python tools/profile_infer.py 16 512 3 > $O/per_layer_on.txt 2>&1
python tools/profile_infer.py 16 512 3 > $O/per_layer_off.txt 2>&1
echo "== $SW"; cat $O/on.json; grep -E "$FLT" $O/per_layer_on.txt
echo "== default"; cat $O/off.json; grep -E "$FLT" $O/per_layer_off.txt
