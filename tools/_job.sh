cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j4
for r in 1 2; do for b in tools/bin/attn_ool0 tools/bin/attn_ool1; do echo "== $b"; timeout 300 $b 30 2>&1 | head -2; FMI_FP8_ONLY=1 timeout 120 $b 30 2>&1 | head -2; done; done > gpurun_out/j4/ool.txt 2>&1
