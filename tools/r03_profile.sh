# Round-3 profiling passes (run on the GPU box through gpurun); keeps only the small summaries under gpurun_out/r03prof/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command (CPU baseline skipped: host-side only)
timeout 560 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
echo "rc1=$?"; cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; tail -3 $O/bench_profiled.err
# 2. FETCH_SIZE pass (own run: counters only beside the kernel trace)
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-prefill --no-serve --no-roofline --no-k-sharded-decode > /dev/null 2> $O/fetch.err
echo "rc2=$?"; python $R/tools/pmc_summary.py /tmp/p_fetch > $O/pmc_FETCH_SIZE.txt; tail -2 $O/fetch.err
# 3. issue / wait counters of the decode kernels
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/p_issue -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-prefill --no-serve --no-roofline --no-k-sharded-decode > /dev/null 2> $O/issue.err
echo "rc3=$?"; python $R/tools/pmc_summary.py /tmp/p_issue > $O/pmc_decode_issue.txt; tail -2 $O/issue.err
cd $R
timeout 150 python tools/gemm3_race_screen.py > $O/race.txt 2>&1; echo "rc4=$?"
timeout 60 python tools/attn_probe.py > $O/attn.txt 2>&1
timeout 60 python tools/serve_kernels.py 7b > $O/serve_kernels.txt 2>&1
ls -la $O; du -sh $O
