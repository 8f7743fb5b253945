#!/bin/bash
# Quick GPU visit: selected tests ($1 = pytest args), bench without CPU baseline, kernel trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest $1 -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
cat $O/bench.json; tail -3 $O/bench.err
export TMPDIR=/tmp
cd /tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; echo "rocprof exit $?"
