cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
export TMPDIR=/tmp
T="tests/test_gpu_engine.py tests/test_gpu_fullsize.py::test_fused_gemv_chain_vs_oracle tests/test_gpu_model_wide.py::test_decode_engine tests/test_gpu_model_depth.py::test_decode_engine_full_depth tests/test_gpu_model.py"
( timeout 600 python -m pytest $T -x -q 2>&1 | tail -8 ) > gpurun_out/r04b/pytest_wgp.log 2>&1
( OB_DEC_WGP=0 timeout 600 python -m pytest $T -x -q 2>&1 | tail -8 ) > gpurun_out/r04b/pytest_nowgp.log 2>&1
( timeout 900 python tools/variant_bench.py run default hb i3 i4 i6 default; echo "--- OB_DEC_WGP=0"; OB_DEC_WGP=0 timeout 300 python tools/variant_bench.py run default ) > gpurun_out/r04b/variants.log 2>&1
( OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so timeout 300 python tools/phase_probe.py ) > gpurun_out/r04b/phase.log 2>&1
tail -3 gpurun_out/r04b/pytest_wgp.log gpurun_out/r04b/pytest_nowgp.log; cat gpurun_out/r04b/variants.log
