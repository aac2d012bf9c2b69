cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r04k/pytest.log 2>&1
tail -n 4 gpurun_out/r04k/pytest.log
bash tools/r04_profile.sh 2>&1 | tail -30
