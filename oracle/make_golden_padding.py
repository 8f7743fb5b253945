"""Generate tests/golden/postp_padding_mask.json by running the UNMODIFIED reference ``Postprocessor`` (CPU) with a
``padding_mask`` (postprocessor.py:58-136: masked frames are filled with -1000 before the max-pool and TRUNCATED before
``nonzero``, so the frame indices behind a masked stretch shift).  TEST INFRASTRUCTURE ONLY; run in the build container:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_padding.py

Inputs are regenerated from the seeds by ``oracle.cases.padding_mask_case`` on both sides; the fixture holds the reference's
outputs only.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BEAT_THIS_REFERENCE", "/root/reference")
sys.path[:0] = [REF, ROOT]

import torch  # noqa: E402

from beat_this.model.postprocessor import Postprocessor  # noqa: E402

from oracle.cases import PADDING_MASK_CASES, padding_mask_case  # noqa: E402


def main():
    pp = Postprocessor("minimal", fps=50)
    out = {}
    for name in PADDING_MASK_CASES:
        beat, down, mask = padding_mask_case(name)
        bt, dt = pp(torch.from_numpy(beat), torch.from_numpy(down), torch.from_numpy(mask))
        if beat.ndim == 1:
            bt, dt = (bt,), (dt,)
        out[name] = {"beats": [[float(x) for x in r] for r in bt], "downbeats": [[float(x) for x in r] for r in dt]}
        print(name, [len(r) for r in bt], [len(r) for r in dt])
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "postp_padding_mask.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
