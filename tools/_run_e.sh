#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 tools/run_attn_bench.sh 20 > gpurun_out/r02e_attn_bench.log 2>&1
