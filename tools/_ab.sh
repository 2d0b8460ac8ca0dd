S="4608,21504,3072;4096,9216,3072;512,9216,3072"
for r in 1 2 3; do for epi in qkv; do for b in base new; do echo "== $epi $b"; FMI_EPI=$epi FMI_SHAPES="$S" ./build/gemm_bench_$b 20 | grep custom | cut -c1-150; done; done; done
