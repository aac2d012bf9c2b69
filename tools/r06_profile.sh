# Round-6 profiling passes (run on the GPU box through gpurun); keeps only the small summaries under gpurun_out/r06prof/.
#   gpurun --timeout 2400 -- 'bash tools/r06_profile.sh'   then copy gpurun_out/r06prof/* to profiles/r06_* (see the end of this file)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command (CPU baseline skipped: host-side only)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
echo "rc1=$?"; cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; tail -2 $O/bench_profiled.err
# 2. FETCH_SIZE pass (own run: counters only beside the kernel trace)
D="--steps 8 --warmup 2 --no-cpu-baseline --no-prefill --no-serve --no-roofline --no-k-sharded-decode --no-eval --no-train"
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python $R/bench.py $D > /dev/null 2> $O/fetch.err
echo "rc2=$?"; python $R/tools/pmc_summary.py /tmp/p_fetch > $O/pmc_FETCH_SIZE.txt; tail -1 $O/fetch.err
# 3. issue / wait counters of the decode kernels
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/p_issue -- python $R/bench.py $D > /dev/null 2> $O/issue.err
echo "rc3=$?"; python $R/tools/pmc_summary.py /tmp/p_issue | grep "ob_dec" > $O/pmc_decode_issue.txt; tail -1 $O/issue.err
# 4. the mixed prefill + decode step (config 5) on 13B shapes: kernel stats of the step, then its HBM traffic
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mixed -- python $R/tools/mixed_probe.py 13b --step-only > $O/mixed_probe_profiled.json 2> $O/mixed.err
echo "rc4=$?"; cp $(find /tmp/p_mixed -name "*kernel_stats.csv" | head -1) $O/mixed_step_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_mixedf -- python $R/tools/mixed_probe.py 13b --step-only > /dev/null 2>> $O/mixed.err
echo "rc5=$?"; python $R/tools/pmc_summary.py /tmp/p_mixedf | grep "ob_" > $O/pmc_mixed_FETCH_SIZE.txt
# 5. key-block decode attention: kernel stats of the long-context probes
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ctx -- python $R/tools/ctx_probe.py > $O/ctx_probe.txt 2> $O/ctx.err
echo "rc6=$?"; cp $(find /tmp/p_ctx -name "*kernel_stats.csv" | head -1) $O/ctx_probe_kernel_stats.csv
cd $R
timeout 200 python tools/serve_ctx_probe.py 7b > $O/serve_ctx_probe.txt 2>&1
timeout 100 python tools/decode_kernels.py > $O/decode_insitu.txt 2>&1
timeout 100 python tools/serve_kernels.py 7b > $O/serve_kernels.txt 2>&1
# 6. model-level parity log (every route's error against the reference's goldens)
rm -f gpurun_out/r06_model_parity.txt
OB_WRITE_PROFILES=1 timeout 900 python -m pytest tests/test_gpu_model_wide.py tests/test_gpu_model_depth.py tests/test_gpu_model_13b_width.py tests/test_gpu_config4.py -q -x > $O/parity_tests.txt 2>&1
cp gpurun_out/r06_model_parity.txt $O/model_parity.txt 2>/dev/null
ls -la $O; du -sh $O
