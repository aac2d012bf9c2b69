cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -k "tile_stats or tp or tensor or sharded or prefill" 2>&1 | tail -6 ) > gpurun_out/r04j/pytest.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-serve --no-eval --no-k-sharded-decode --no-roofline ) > gpurun_out/r04j/bench.json 2> gpurun_out/r04j/bench.err
tail -n 4 gpurun_out/r04j/pytest.log; tail -n 3 gpurun_out/r04j/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04j/bench.json').read().strip().splitlines()[-1])
p=d['prefill_k_sharded']; print('k_sharded',p['TFLOPs'],'n_sharded',p['n_sharded'].get('TFLOPs'),p['n_sharded'].get('ms_per_call'),'token',p['token_sharded']['TFLOPs'])
print('prefill_model',d['prefill_model']['ms'],'tp',d['prefill_model_tp']['ms'])
PY
