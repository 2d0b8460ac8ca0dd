cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j7
python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/j7/gpu_tests.txt 2>&1; echo "rc $?" >> gpurun_out/j7/gpu_tests.txt
python bench.py > gpurun_out/j7/bench.json 2> gpurun_out/j7/bench.err; echo "rc $?" >> gpurun_out/j7/bench.err
bash tools/profile_round.sh r05 > gpurun_out/j7/profile.log 2>&1
