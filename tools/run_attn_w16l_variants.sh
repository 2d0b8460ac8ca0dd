#!/bin/bash
# Variants of attention_w16l's generated stream, built HERE (hipcc cross-compiles; tools/bin/ travels with gpurun, build/ does not):
#   tools/run_attn_w16l_variants.sh "tag:ENV=val,ENV=val" ...     e.g.  "la20:AW16L_LOOKAHEAD=20"  "novalu:AW16L_X=novalu"  "ey10:AW16L_EY=10"
# then on the GPU box:  for b in tools/bin/attn_w16l_*; do echo == $b; $b 30 | grep "w16l vs w16" | head -3; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build tools/bin
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  ( env $(echo "$envs" | tr ',' ' ') AW16L_TAG=$tag python3 tools/gen_attention_w16l.py 2> /dev/null
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1 -Ibuild \
      -DFMI_AW16L_LOOP_INC="\"../../build/attention_w16l_loop_$tag.inc\"" tools/attn_bench.hip -o tools/bin/attn_w16l_$tag ) &
  while [ "$(jobs -r | wc -l)" -ge 7 ]; do sleep 1; done
done
wait
ls tools/bin/
