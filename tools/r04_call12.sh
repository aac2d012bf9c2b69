cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
export TMPDIR=/tmp
T="tests/test_gpu_engine.py tests/test_gpu_model_wide.py tests/test_gpu_model_depth.py tests/test_gpu_serving.py tests/test_gpu_model.py"
( timeout 900 python -m pytest $T -x -q 2>&1 | tail -6 ) > gpurun_out/r04l/pytest.log 2>&1
( timeout 300 python tools/decode_kernels.py ) > gpurun_out/r04l/insitu.log 2>&1
( timeout 200 python tools/serve_kernels.py 7b ) > gpurun_out/r04l/serve_kernels.log 2>&1
( timeout 300 python tools/variant_bench.py run default ) > gpurun_out/r04l/variants.log 2>&1
tail -n 3 gpurun_out/r04l/pytest.log; cat gpurun_out/r04l/variants.log; grep -v Warn gpurun_out/r04l/insitu.log | tail -n 11; grep "attn\|per step" gpurun_out/r04l/serve_kernels.log
