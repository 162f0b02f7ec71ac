#!/bin/bash
# sample the shader clock while one GEMM shape runs in a loop
python tools/bench_igemm.py --iters 6000 --only fwd --shape ${1:-up2.c0} > gpurun_out/clk_run.log 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1; sleep 0.3; done
wait $PID
tail -2 gpurun_out/clk_run.log
