#!/bin/bash
# GPU box: fingerprint of the division-variant, CU trace and a short bench of the current tree; $1 = tag
T=${1:-x}
VF_LIBRARY=build/ab/gatediv.so timeout 300 python tools/fingerprint.py gatediv > gpurun_out/fingerprint_gatediv.txt 2>&1
VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_cu.py 200 > gpurun_out/r3_trace_200_$T.txt 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/r3_bench_$T.json 2>gpurun_out/r3_bench_$T.err
python tools/bench_line.py gpurun_out/r3_bench_$T.json
head -32 gpurun_out/r3_trace_200_$T.txt
