#!/bin/bash
# Variants of the all-e4m3 stream (AW16L_MODE=fp8pv), built HERE (tools/bin/ travels with gpurun):
#   tools/run_attn_pv8_variants.sh "tag:ENV=val,ENV=val" ...   e.g. "ex4:AW16L_F8_EX=4"  "novalu:AW16L_X=novalu"
# then on the GPU box:  for b in tools/bin/attn_pv8_*; do echo == $b; FMI_FP8_ONLY=1 $b 30 | head -2; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build tools/bin
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  [ "$envs" = "$spec" ] && envs=""
  ( env $(echo "$envs" | tr ',' ' ') AW16L_MODE=fp8pv AW16L_TAG=$tag python3 tools/gen_attention_w16l.py 2> /dev/null
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1 -Ibuild \
      -DFMI_AW16LF8PV_LOOP_INC="\"../../build/attention_w16lf8pv_loop_$tag.inc\"" tools/attn_bench.hip -o tools/bin/attn_pv8_$tag ) &
  while [ "$(jobs -r | wc -l)" -ge 7 ]; do sleep 1; done
done
wait
ls tools/bin/
