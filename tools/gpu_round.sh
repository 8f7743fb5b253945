#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel trace. Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
cat $O/bench.json
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; echo "rocprof exit $?"
ls -R $O/prof | head -30
