#!/bin/bash
# Tuning aid: samples sclk / power with rocm-smi while a pipelined and a serialised bench run.
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clocks.log &
W=$!
python bench.py --steps 40 --warmup 2 --no-cpu-baseline > gpurun_out/cw_pipe.log 2>&1
echo "--- serial" >> gpurun_out/clocks.log
RA_PIPELINE_CHUNKS=1 python bench.py --steps 30 --warmup 2 --no-cpu-baseline > gpurun_out/cw_serial.log 2>&1
wait $W
grep -o '"ms_per_step[^,]*' gpurun_out/cw_pipe.log gpurun_out/cw_serial.log
