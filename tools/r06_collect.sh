# Copies the summaries of tools/r06_profile.sh (gpurun_out/r06prof) and of the plain bench run (gpurun_out/r06/bench.json) into profiles/r06_*
# and restamps profiles/pmc_traffic.json for the current kernel sources.  Run in the repo root after the gpurun call returned.
set -e
O=gpurun_out/r06prof
cp $O/bench_kernel_stats.csv profiles/r06_bench_rocprofv3_kernel_stats.csv
cp $O/bench_profiled.json profiles/r06_bench_profiled_run.json
cp $O/pmc_FETCH_SIZE.txt profiles/r06_pmc_FETCH_SIZE.txt
cp $O/pmc_decode_issue.txt profiles/r06_pmc_decode_issue.txt
cp $O/mixed_step_kernel_stats.csv profiles/r06_mixed_step_rocprofv3_kernel_stats.csv
cp $O/mixed_probe_profiled.json profiles/r06_mixed_probe_profiled_run.json
cp $O/pmc_mixed_FETCH_SIZE.txt profiles/r06_pmc_mixed_step_FETCH_SIZE.txt
python tools/kstats.py $O/ctx_probe_kernel_stats.csv > profiles/r06_ctx_probe_kernels.txt
python tools/kstats.py $O/mixed_step_kernel_stats.csv > profiles/r06_mixed_step_kernels.txt
cp $O/serve_ctx_probe.txt profiles/r06_serve_ctx_probe.txt
cp $O/decode_insitu.txt profiles/r06_decode_insitu.txt
cp $O/serve_kernels.txt profiles/r06_serve_step_kernels.txt
cp $O/model_parity.txt profiles/r06_model_parity.txt
[ -s gpurun_out/r06/bench.json ] && cp gpurun_out/r06/bench.json profiles/r06_bench.json
python tools/make_pmc_traffic.py $O/pmc_FETCH_SIZE.txt | tail -1
python -c "
import json,sys; sys.path.insert(0,'.'); import bench; d=json.load(open('profiles/pmc_traffic.json')); print('stamp matches the sources:', d.get('csrc_sha')==bench.csrc_sha())"
