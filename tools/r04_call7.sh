cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
export TMPDIR=/tmp
S=$(date +%s)
( timeout 1500 python bench.py ) > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err
echo "bench wall: $(( $(date +%s) - S )) s"
tail -n 5 gpurun_out/r04g/bench.err; tail -c 7000 gpurun_out/r04g/bench.json
