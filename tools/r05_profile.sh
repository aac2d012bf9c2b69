# Round-5 profiling passes (run on the GPU box through gpurun); keeps only the small summaries under gpurun_out/r05prof/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command (CPU baseline skipped: host-side only)
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
echo "rc1=$?"; cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; tail -2 $O/bench_profiled.err
# 2. FETCH_SIZE pass (own run: counters only beside the kernel trace)
D="--steps 8 --warmup 2 --no-cpu-baseline --no-prefill --no-serve --no-roofline --no-k-sharded-decode --no-eval --no-train"
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python $R/bench.py $D > /dev/null 2> $O/fetch.err
echo "rc2=$?"; python $R/tools/pmc_summary.py /tmp/p_fetch > $O/pmc_FETCH_SIZE.txt; tail -1 $O/fetch.err
# 3. issue / wait counters of the decode kernels
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/p_issue -- python $R/bench.py $D > /dev/null 2> $O/issue.err
echo "rc3=$?"; python $R/tools/pmc_summary.py /tmp/p_issue | grep "ob_dec" > $O/pmc_decode_issue.txt; tail -1 $O/issue.err
if [ -z "$R05_SHORT" ]; then
# 4. prefill layer [16384, 4096] -> 11008: kernel stats, then the MFMA counters of the same command
OB_ONE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_pl -- python $R/tools/prefill_probe.py > $O/prefill_layer.txt 2> $O/prefill_layer.err
echo "rc4=$?"; cp $(find /tmp/p_pl -name "*kernel_stats.csv" | head -1) $O/prefill_layer_kernel_stats.csv
OB_ONE=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_pm -- python $R/tools/prefill_probe.py > /dev/null 2> $O/prefill_mfma.err
echo "rc5=$?"; python $R/tools/pmc_summary.py /tmp/p_pm | grep "gemm3\|scale_rows\|layernorm" > $O/pmc_prefill_mfma.txt
OB_ONE=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/p_pm2 -- python $R/tools/prefill_probe.py > /dev/null 2>> $O/prefill_mfma.err
echo "rc6=$?"; python $R/tools/pmc_summary.py /tmp/p_pm2 | grep "gemm3" >> $O/pmc_prefill_mfma.txt
fi
cd $R
timeout 100 python tools/decode_kernels.py > $O/decode_insitu.txt 2>&1
timeout 100 python tools/serve_kernels.py 7b > $O/serve_kernels.txt 2>&1
# 6. the persistent-token skeleton (tools/persist_probe.hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/persist_probe tools/persist_probe.hip && timeout 150 /tmp/persist_probe > $O/persist_skeleton.txt 2>&1
ls -la $O; du -sh $O
