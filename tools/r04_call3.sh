cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
export TMPDIR=/tmp
T="tests/test_gpu_engine.py tests/test_gpu_fullsize.py::test_fused_gemv_chain_vs_oracle tests/test_gpu_model_wide.py::test_decode_engine tests/test_gpu_model_depth.py::test_decode_engine_full_depth tests/test_gpu_model.py tests/test_gpu_config4.py"
( timeout 600 python -m pytest $T -x -q 2>&1 | tail -8 ) > gpurun_out/r04c/pytest_wgp.log 2>&1
( OB_DEC_WGP=0 timeout 600 python -m pytest $T -x -q 2>&1 | tail -8 ) > gpurun_out/r04c/pytest_nowgp.log 2>&1
( timeout 900 python tools/variant_bench.py run default hb i3 i4 hbi4 default; echo "--- OB_DEC_WGP=0"; OB_DEC_WGP=0 timeout 300 python tools/variant_bench.py run default ) > gpurun_out/r04c/variants.log 2>&1
( timeout 300 python tools/decode_kernels.py ) > gpurun_out/r04c/insitu.log 2>&1
( OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so timeout 300 python tools/phase_probe.py ) > gpurun_out/r04c/phase.log 2>&1
for f in pytest_wgp pytest_nowgp; do tail -n 3 gpurun_out/r04c/$f.log; done; cat gpurun_out/r04c/variants.log; grep -v Warn gpurun_out/r04c/insitu.log | tail -n 16
