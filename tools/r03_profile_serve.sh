# PMC pass over the 32-slot batched step (7B): issue / wait / VALU / MFMA counters of the LDS-DMA skinny GEMM launches.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OB_STEADY_ONLY=1 timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/p_serve -- python $R/tools/serve_probe.py 7b > $O/serve_pmc.out 2> $O/serve_pmc.err
echo "rc=$?"; python $R/tools/pmc_summary.py /tmp/p_serve | grep "skinny3\|b_norm\|swiglu\|attn" > $O/pmc_serve_step.txt; tail -2 $O/serve_pmc.err
OB_STEADY_ONLY=1 timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d /tmp/p_serve2 -- python $R/tools/serve_probe.py 7b > /dev/null 2> $O/serve_pmc2.err
echo "rc=$?"; python $R/tools/pmc_summary.py /tmp/p_serve2 | grep "skinny3" >> $O/pmc_serve_step.txt; tail -2 $O/serve_pmc2.err
wc -l $O/pmc_serve_step.txt
