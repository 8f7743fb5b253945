"""bench.py's N > 1 code path on a box with ONE GPU: `--share-gpu` puts every rank on cuda:0 and the
collectives on gloo, so the self-launch under torch.distributed.run, the sharded workload, the gathers, the max-over-ranks timing,
the per-rank breakdowns AND the failure rule (a rank that raises must take the job down at once instead of leaving the others in a
collective until a watchdog fires) run for real.  Not a measurement."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1", "--min-seconds", "0",
         "--tracks", "2", "--no-cpu-baseline", "--watchdog", "240"]


@pytest.mark.timeout(420)
def test_two_ranks_on_one_gpu_run_the_distributed_bench():
    r = subprocess.run(BENCH, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak"
    assert len(d["timed_region"]["per_rank_ms_per_step"]) == 2
    assert d["config"]["global_chunks"] == 2 * d["config"]["chunks_per_gpu"]
    s = d["strong_scaling_cfg4"]
    assert s["global_chunks"] == 512 and s["chunks_per_gpu"] == 256 and len(s["per_rank_ms"]) == 2 and len(s["gather_ms"]) == 2
    assert d["value"] > 0 and "share-gpu" in d["rccl_note"]


@pytest.mark.timeout(300)
def test_a_rank_that_raises_takes_the_job_down_at_once():
    t0 = time.time()
    r = subprocess.run(BENCH + ["--no-extras", "--fail-rank", "1", "--fail-step", "1"], capture_output=True, text=True, timeout=280, cwd=ROOT)
    took = time.time() - t0
    assert r.returncode != 0
    assert "aborting the job" in r.stderr and "--fail-rank 1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no JSON line from a failed job
    assert took < 150, f"the surviving rank waited {took:.0f} s (collective timeout is 180 s, the watchdog 240 s)"


def test_more_gpus_than_the_node_has_fails_fast_with_a_clear_message():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 64 but this node shows" in (r.stderr + r.stdout)
