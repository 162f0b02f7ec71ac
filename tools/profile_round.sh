set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-p}          # output suffix: gpurun_out/r3${S}_{a,b,f,w}
B="python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3${S}_a -o a -- $B --sync-wgrad > $R/gpurun_out/r3${S}_a.log 2>&1
echo "a done"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3${S}_b -o b -- $B --no-roofline > $R/gpurun_out/r3${S}_b.log 2>&1
echo "b done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3${S}_f -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --sync-wgrad > $R/gpurun_out/r3${S}_f.log 2>&1
echo "f done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3${S}_w -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --sync-wgrad > $R/gpurun_out/r3${S}_w.log 2>&1
echo "w done"
cd $R
ls gpurun_out/r3${S}_a gpurun_out/r3${S}_f | head
tail -c 600 gpurun_out/r3${S}_a.log
