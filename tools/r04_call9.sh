cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_serving.py -x -q -k bursts 2>&1 | tail -3 ) > gpurun_out/r04i/pytest.log 2>&1
( timeout 400 python tools/variant_bench.py run default; echo "--- OB_DEC_I8_SINGLE=1"; OB_DEC_I8_SINGLE=1 timeout 300 python tools/variant_bench.py run default; OB_DEC_I8_SINGLE=1 timeout 300 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py::test_fused_gemv_chain_vs_oracle -x -q 2>&1 | tail -3 ) > gpurun_out/r04i/variants.log 2>&1
tail -n 2 gpurun_out/r04i/pytest.log; cat gpurun_out/r04i/variants.log
