#!/bin/bash
# Timing experiments on the generated stream of attention_w16_kernel / attention_w32_kernel (wrong results by construction):
# builds one attn_bench per variant HERE (no GPU needed); run them on the GPU box with
#   for v in ...; do build/attn_bench_${K}_$v 20 | grep -B2 "L=4608"; done
# K=w16|w32 VARIANTS="novalu nomfma ..." tools/run_attn_w16_ablations.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p build
K=${K:-w16}
UP=$(echo $K | tr a-z A-Z)
for v in ${VARIANTS:-novalu noexp nodma nobarrier halfreads nomfma nowait}; do
  env A${UP}_X=$v python3 tools/gen_attention_$K.py
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1 \
    -DFMI_A${UP}_LOOP_INC="\"../../build/attention_${K}_loop_$v.inc\"" tools/attn_bench.hip -o build/attn_bench_${K}_$v &
done
wait
