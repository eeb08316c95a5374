#!/bin/bash
# Development tool (GPU box): the rocprofv3 passes behind profiles/round6_*, profiles/decode_fp8_pmc.json and
# profiles/moe_tiled_gemm_pmc_r6.json.  Every pass runs under its own timeout; counters are collected in their own
# runs (--pmc with --kernel-trace only).  tools/round6_summarise.py condenses the outputs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6prof; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python $R/bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-low-latency > $O/bench.log 2>&1; echo "bench rc=$?"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/pmc_decode.py mixed > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/pmc_decode.py mixed > $O/pmc_write.log 2>&1; echo "write rc=$?"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python $R/tools/pmc_decode.py mixed > $O/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_moe -- python $R/tools/pmc_moe.py > $O/pmc_moe.log 2>&1; echo "moe rc=$?"
timeout 400 python $R/tools/moe_clock.py $O/round6_moe_clock.json > $O/moe_clock.log 2>&1; echo "clock rc=$?"
grep -h "^{" $O/bench.log | tail -1 | cut -c1-600
# keep the merge-back small: drop the per-dispatch traces of the bench pass
find $O/bench -name "*kernel_trace.csv" -delete
du -sh $O
