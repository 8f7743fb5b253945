#!/bin/bash
# round 6: bench.py's 8-rank code path on a ONE-GPU box (--share-gpu: every rank on cuda:0, collectives on gloo) launched the way the
# driver launches N > 1, with the default workload.  Evidence that the path executes at world 8 -- not a scaling measurement.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r06world8
mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 \
  --share-gpu --steps 6 --warmup 2 --min-seconds 0 --no-cpu-baseline 2>$O/bench.err > $O/bench.json
echo "exit $?"; grep -v "amdgpu.ids\|socket.cpp" $O/bench.err | tail -12
python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling", "rccl_ranks", "rccl_note")})
print("timed_region", json.dumps(d["timed_region"]))
print("strong", json.dumps(d["strong_scaling_cfg4"])[:900])
PY
