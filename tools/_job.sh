cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j6
for i in 1 2; do python tools/hipblaslt_yardstick.py 2>&1 | grep -v "^W2026\|amdgpu.ids" ; done > gpurun_out/j6/yard2.txt 2>&1
