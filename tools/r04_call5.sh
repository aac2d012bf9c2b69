cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
export TMPDIR=/tmp
T="tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_model_wide.py tests/test_gpu_model_depth.py tests/test_gpu_serving.py tests/test_gpu_train.py"
( timeout 900 python -m pytest $T -x -q 2>&1 | tail -8 ) > gpurun_out/r04e/pytest.log 2>&1
( timeout 600 python tools/variant_bench.py run --serve default default ) > gpurun_out/r04e/variants.log 2>&1
( timeout 300 python tools/decode_kernels.py ) > gpurun_out/r04e/insitu.log 2>&1
( OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so timeout 300 python tools/phase_probe.py ) > gpurun_out/r04e/phase.log 2>&1
tail -n 4 gpurun_out/r04e/pytest.log; cat gpurun_out/r04e/variants.log; grep -v Warn gpurun_out/r04e/insitu.log | tail -n 12
