"""Does the package power cap apply from inside this container, and what does the headline do under a lower cap?
(round 4's `perfdeterminism` sweep did not apply and said nothing.)  Development tool.

    python tools/cap_sweep.py [watts ...]      default 1200 1000 800

For every requested cap: set it (librocm_smi64 rsmi_dev_power_cap_set, then `rocm-smi --setpoweroverdrive` as a second
attempt), READ IT BACK, and only if the read-back equals the request run a short bench (no extras) and record ms per step,
joules per step and the average power from the energy accumulator.  The refusing command's output is recorded otherwise.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.smi import Smi


def bench(extra=()):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--no-dist", "--steps", "20",
                        "--warmup", "3", *extra], capture_output=True, text=True, timeout=600)
    try:
        d = json.loads(p.stdout.strip().splitlines()[-1])
        return {"ms_per_step": d["ms_per_step"], "value": d["value"], "energy": d.get("energy")}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e), "stderr_tail": p.stderr[-400:]}


def main():
    smi = Smi()
    out = {"cap_now_W": smi.cap_w(), "cap_range_W": smi.cap_range_w(), "sweep": []}
    print(json.dumps(out), flush=True)
    base = smi.cap_w()
    out["baseline"] = bench()
    print("baseline", json.dumps(out["baseline"]), flush=True)
    for w in [float(a) for a in sys.argv[1:]] or [1200.0, 1000.0, 800.0]:
        row = {"requested_W": w}
        row["rsmi_status"] = smi.cap_set_w(w)
        row["readback_W"] = smi.cap_w()
        if row["readback_W"] != w:
            p = subprocess.run(["rocm-smi", "--setpoweroverdrive", str(int(w)), "--autorespond", "y"], capture_output=True, text=True)
            row["rocm_smi_rc"] = p.returncode
            row["rocm_smi_output"] = (p.stdout + p.stderr)[-600:]
            row["readback_W"] = smi.cap_w()
        row["applied"] = row["readback_W"] == w
        if row["applied"]:
            row["bench"] = bench()
        out["sweep"].append(row)
        print(json.dumps(row), flush=True)
    if base is not None and smi.cap_w() != base:
        smi.cap_set_w(base)
    out["cap_restored_W"] = smi.cap_w()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cap_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
