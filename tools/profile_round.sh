set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3p_a -o a -- $B --sync-wgrad > $R/gpurun_out/r3p_a.log 2>&1
echo "a done"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3p_b -o b -- $B --no-roofline > $R/gpurun_out/r3p_b.log 2>&1
echo "b done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3p_f -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --sync-wgrad > $R/gpurun_out/r3p_f.log 2>&1
echo "f done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3p_w -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --sync-wgrad > $R/gpurun_out/r3p_w.log 2>&1
echo "w done"
cd $R
ls gpurun_out/r3p_a gpurun_out/r3p_f | head
tail -c 600 gpurun_out/r3p_a.log
