cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
export TMPDIR=/tmp
T="tests/test_gpu_engine.py tests/test_gpu_fullsize.py::test_fused_gemv_chain_vs_oracle tests/test_gpu_model_wide.py::test_decode_engine tests/test_gpu_model_depth.py::test_decode_engine_full_depth"
( timeout 900 python -m pytest $T -x -q 2>&1 | tail -8 ) > gpurun_out/r04f/pytest.log 2>&1
( timeout 600 python tools/variant_bench.py run default ) > gpurun_out/r04f/variants.log 2>&1
( timeout 300 python tools/decode_kernels.py ) > gpurun_out/r04f/insitu.log 2>&1
( timeout 200 python tools/serve_kernels.py 7b ) > gpurun_out/r04f/serve_kernels.log 2>&1
( OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so timeout 300 python tools/skinny3_phase_probe.py ) > gpurun_out/r04f/sk3_phase.log 2>&1
tail -n 4 gpurun_out/r04f/pytest.log; cat gpurun_out/r04f/variants.log; grep -v Warn gpurun_out/r04f/insitu.log | tail -n 11; grep -v Warn gpurun_out/r04f/serve_kernels.log | tail -n 16
