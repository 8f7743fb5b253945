#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/pmc1 $O/pmc2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc1 -o pmc --output-format csv -- python $R/tools/attn_probe.py > $O/pmc1.log 2>&1; echo "pmc1 exit $?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc2 -o pmc --output-format csv -- python $R/tools/attn_probe.py > $O/pmc2.log 2>&1; echo "pmc2 exit $?"
ls -R $O/pmc1 | head; tail -3 $O/pmc1.log
