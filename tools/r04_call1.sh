cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r04a/pytest.log 2>&1
( timeout 60 tools/kernarg_probe; echo ---- preload build; timeout 60 tools/kernarg_probe_pl ) > gpurun_out/r04a/kernarg.log 2>&1
( timeout 600 python tools/chain_probe.py 7b 13b ) > gpurun_out/r04a/chains.log 2>&1
( OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so timeout 300 python tools/phase_probe.py ) > gpurun_out/r04a/phase.log 2>&1
tail -3 gpurun_out/r04a/pytest.log; cat gpurun_out/r04a/kernarg.log; cat gpurun_out/r04a/chains.log
