# round 5: clock / power probe, WN layer kernel A/B (working tree against csrc/libsvoc_hip_ab.so), 1 x 200 kernel timeline under graph replay
cd /root/repo
O=gpurun_out/${1:-r05d}; mkdir -p $O
OLD=/root/repo/smart-vocoder_amd/csrc/libsvoc_hip_ab.so
timeout 300 tools/clock_power_probe > $O/clock_power_probe.txt 2>&1
for i in 1 2; do
  python tools/wn_timeline.py 16 512 > $O/wn_new_$i.txt 2>&1
  SVOC_LIB=$OLD python tools/wn_timeline.py 16 512 > $O/wn_old_$i.txt 2>&1
  python tools/step_ab.py >> $O/step_new.json 2>> $O/ab.err
  SVOC_LIB=$OLD python tools/step_ab.py >> $O/step_old.json 2>> $O/ab.err
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt200 --output-format csv -- python /root/repo/tools/small_shape_run.py 1 200 20 > /root/repo/$O/small_run.txt 2>&1; python /root/repo/tools/timeline.py /tmp/kt200 400 > /root/repo/$O/kernel_timeline_1x200_graph.txt 2>&1 )
cat $O/clock_power_probe.txt; head -3 $O/wn_new_1.txt $O/wn_old_1.txt $O/wn_new_2.txt $O/wn_old_2.txt; cat $O/step_new.json $O/step_old.json; head -5 $O/kernel_timeline_1x200_graph.txt
