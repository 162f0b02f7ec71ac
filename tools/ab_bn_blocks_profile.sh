# Same-box A/B of UCLSTM_BN_BWD_BLOCKS inside the serialised step: rocprofv3 --kernel-trace --stats, alternating settings.
# (the variable is exported in this shell; the program after `--` is python3 itself)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for b in 4096 1024; do
    export UCLSTM_BN_BWD_BLOCKS=$b
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3ab_${b}_$i -o s -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-secondary --sync-wgrad > $R/gpurun_out/r3ab_${b}_$i.log 2>&1
    echo "blocks $b run $i done"
  done
done
