#!/bin/bash
# Sweep of the fragment look-ahead of attention_w16's generated stream (AW16_LOOKAHEAD: a fragment is read that many MFMA slots ahead of its
# first use; 12 is the largest value the 8 fragment buffers allow under rule 3 and the committed one).  Builds one attn_bench per value on the
# GPU box (build/ does not travel) and prints the L = 4608 lines, twice in alternation.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
for la in ${LAS:-6 8 10 12}; do
  AW16_X=la$la AW16_LOOKAHEAD=$la python3 tools/gen_attention_w16.py 2> /dev/null
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result -DFMI_ALT_KERNELS=1 \
    -DFMI_AW16_LOOP_INC="\"../../build/attention_w16_loop_la$la.inc\"" tools/attn_bench.hip -o build/attn_bench_w16_la$la &
done
wait
for rep in 1 2; do
  for la in ${LAS:-6 8 10 12}; do
    echo "== lookahead $la"
    ./build/attn_bench_w16_la$la 30 | grep -B2 "L=4608" | grep -E "w16 vs pp|B=1 H=24 L=4608|B=2 H=24 L=4608" | cut -c1-200
  done
done
