cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_serving.py tests/test_gpu_model_wide.py tests/test_gpu_fullsize.py -x -q -k "serving or batch or slots or continuous" 2>&1 | tail -6 ) > gpurun_out/r04h/pytest.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-prefill --no-eval --no-k-sharded-decode --no-roofline ) > gpurun_out/r04h/bench.json 2> gpurun_out/r04h/bench.err
tail -n 3 gpurun_out/r04h/pytest.log; tail -n 3 gpurun_out/r04h/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04h/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['continuous_batch'])
PY
