cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j8
python -m pytest tests/test_gpu_vae.py tests/test_gpu_seam.py tests/test_gpu_pipeline.py tests/test_gpu_ops.py "tests/test_gpu_production_shapes.py::test_vae_decode_flux_config_matches_oracle" "tests/test_gpu_production_shapes.py::test_vae_attn_block_4096_tokens_matches_oracle" "tests/test_gpu_production_shapes.py::test_conv2d_cin512_matches_oracle" tests/test_gpu_fullsize.py tests/test_gpu_shared_device.py -x -q -s -m gpu > gpurun_out/j8/tests.txt 2>&1; echo "rc $?" >> gpurun_out/j8/tests.txt
python tools/vae_bench.py > gpurun_out/j8/vae_bench.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/j8/smoke.txt 2>&1; echo "rc $?" >> gpurun_out/j8/smoke.txt
