#!/bin/bash
# two PMC passes + kernel trace of tools/kprof.py; summaries under gpurun_out/pmc_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/tools/kprof.py ${1:-8192} ${2:-8192} ${3:-2} 2 > $R/gpurun_out/prof_trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/prof_pmcA -- python3 $R/tools/kprof.py ${1:-8192} ${2:-8192} ${3:-2} 1 > $R/gpurun_out/prof_pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/prof_pmcB -- python3 $R/tools/kprof.py ${1:-8192} ${2:-8192} ${3:-2} 1 > $R/gpurun_out/prof_pmcB.log 2>&1
find $R/gpurun_out/prof_trace $R/gpurun_out/prof_pmcA $R/gpurun_out/prof_pmcB -name "*.csv" | head -20
