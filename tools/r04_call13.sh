cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
export TMPDIR=/tmp
( timeout 900 python bench.py ) > gpurun_out/r04m/bench.json 2> gpurun_out/r04m/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04m/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:40], d['roofline']['frac'], [ (k['kernel'],k['avg_launch_us']) for k in d['roofline']['per_kernel']])
print(d['continuous_batch']['ms_per_step'], d['decode_k_sharded']['single_gpu_engine'], d['prefill_model']['ms'], d['prefill_model_tp']['ms'])
PY
R04_SHORT=1 bash tools/r04_profile.sh 2>&1 | tail -12
( OB_LIB=onebit_amd/csrc/variants/libonebit_stamps.so timeout 300 python tools/phase_probe.py ) > gpurun_out/r04prof/phase.txt 2>&1
