# Same-box A/B of one environment switch inside the serialised step: rocprofv3 --kernel-trace --stats, alternating values
#   bash tools/ab_env_profile.sh VAR tag value1 value2 [value3 ...]      (the variable is exported in this shell; the program after `--` is python3)
set -e
VAR=$1; T=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for v in "$@"; do
    export $VAR=$v
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3ab_${T}_${v}_$i -o s -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-secondary --sync-wgrad > $R/gpurun_out/r3ab_${T}_${v}_$i.log 2>&1
    echo "$VAR=$v run $i done"
  done
done
