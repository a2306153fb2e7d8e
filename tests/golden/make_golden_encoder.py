"""Golden vectors of the variant and polish summary encoders, produced by the REFERENCE's C++ compiled into
oracle/_ref (build container only):   python tests/golden/make_golden_encoder.py
Stores only outputs; inputs are regenerated from seeds by tests/test_encoder_oracle.py::_case."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import pileup_utils as pu                    # noqa: E402
from test_encoder_oracle import CASES, POLISH_CASES, _case, _polish_case  # noqa: E402

ref = pu.load_reference_encoder()
assert ref is not None, "needs /root/reference"
for name in sorted(CASES)[:4]:
    pile, params = _case(**CASES[name])
    out = pu.run_variant(ref, pile, params, reference_impl=True)
    np.savez_compressed(os.path.join(HERE, f"encoder_variant_{name}.npz"), positions=out["positions"],
                        depths=out["depths"], candidate_frequency=out["candidate_frequency"],
                        images=out["images"].astype(np.int16), candidates="\n".join(out["candidates"]))
    print(name, len(out["candidates"]), "candidates")

pref = pu.load_reference_polish_encoder()
assert pref is not None, "needs /root/reference"
for name in sorted(POLISH_CASES)[:3]:
    pile, start, end = _polish_case(**POLISH_CASES[name])
    img, pos = pu.run_polish_reference(pref, pile, start, end)
    np.savez_compressed(os.path.join(HERE, f"encoder_polish_{name}.npz"), image=img, positions=pos)
    print("polish", name, len(img), "rows")
