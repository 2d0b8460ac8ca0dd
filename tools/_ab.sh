S="4608,21504,3072;4096,12288,3072;4096,9216,3072;4608,3072,3072"
for r in 1 2; do for epi in store gelu qkv resid; do for b in base new; do echo "== $epi $b"; FMI_EPI=$epi FMI_SHAPES="$S" ./build/gemm_bench_$b 20 | grep custom | cut -c1-150; done; done; done
