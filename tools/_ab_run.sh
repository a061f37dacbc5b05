export SVOC_FUSE=0
for rep in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/tools/libsvoc_old.so" "$GRAFT_REPO_ROOT/tools/libsvoc_oldloop.so"; do
  for cfg in "128 3 1 32768" "128 11 1 32768"; do
    SVOC_LIB=$lib timeout 120 python tools/conv_probe.py $cfg 16 20 2>&1 | grep -E "^#" | sed "s#^#lib=${lib##*/} #"
  done
done
done
