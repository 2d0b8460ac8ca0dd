set -e
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $F tools/gemm_bench.hip -o build/gemm_bench
for epi in store gelu resid; do
  FMI_EPI=$epi FMI_SHAPES="4608,21504,1024;4608,21504,2048;4608,21504,3072;4608,21504,4096;4608,21504,6144;4608,3072,3072;4608,3072,6144;4608,3072,15360" ./build/gemm_bench 20 | grep -v "^pads\|^weight"
done
