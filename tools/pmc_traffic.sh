#!/bin/bash
# HBM traffic counters (separate passes, MI355X_MICROARCH.md HBM section) for every kernel of the bench workload.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf $O/pmc_fetch $O/pmc_write
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1; echo "fetch exit $?"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1; echo "write exit $?"
