#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/pmcg1 $O/pmcg2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmcg1 -o pmc --output-format csv -- python $R/tools/gemm3_probe.py ff1 ff2 > $O/pmcg1.log 2>&1; echo "pmc1 exit $?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC -d $O/pmcg2 -o pmc --output-format csv -- python $R/tools/gemm3_probe.py ff1 ff2 > $O/pmcg2.log 2>&1; echo "pmc2 exit $?"
