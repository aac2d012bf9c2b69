cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r04d/pytest.log 2>&1
( timeout 900 python tools/variant_bench.py run default nohb sw0 sw1 default; echo "--- OB_DEC_SWIGLU_WGS=128"; OB_DEC_SWIGLU_WGS=128 timeout 300 python tools/variant_bench.py run default ) > gpurun_out/r04d/variants.log 2>&1
( timeout 300 python tools/decode_kernels.py ) > gpurun_out/r04d/insitu.log 2>&1
( OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so timeout 300 python tools/phase_probe.py ) > gpurun_out/r04d/phase.log 2>&1
tail -n 4 gpurun_out/r04d/pytest.log; cat gpurun_out/r04d/variants.log; grep -v Warn gpurun_out/r04d/insitu.log | tail -n 14
