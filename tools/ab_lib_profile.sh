# Same-box A/B of two builds of the library inside the serialised step: rocprofv3 --kernel-trace --stats, alternating
#   bash tools/ab_lib_profile.sh tools/probes/_bin/libuclstm_old.so [tag]
# "old" = the library given (UCLSTM_LIB, exported in this shell; the program after `--` is python3 itself), "new" = the tree's own.
set -e
OLD=$GRAFT_REPO_ROOT/$1
T=${2:-lib}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export UCLSTM_LIB=$OLD; else unset UCLSTM_LIB; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3ab_${T}_${v}_$i -o s -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-secondary --sync-wgrad > $R/gpurun_out/r3ab_${T}_${v}_$i.log 2>&1
    echo "$v run $i done"
  done
done
