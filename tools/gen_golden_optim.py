#!/usr/bin/env python
"""Freeze golden vectors of the REFERENCE optimizers (build container only): imports AdamW / QHM from
/root/reference/src/optim, steps them on the seeded cases of tests/optim_cases.py, asserts that oracle/optim_oracle.py
is bit-identical, and writes the reference's parameters after steps 1 and 5 to tests/golden/optim_reference.npz
(outputs only; inputs are regenerated from the seed).

    python tools/gen_golden_optim.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/src"


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present: golden vectors can only be generated in the build container")
    sys.path.insert(0, REF)
    from optim.optimization import AdamW as RefAdamW
    from optim.qhm import QHM as RefQHM
    from oracle import optim_oracle as OO
    import optim_cases as OC

    def ref_opt(kind, groups, hyper):
        return RefAdamW(groups, **hyper) if kind == "adamw" else RefQHM(groups, **hyper)

    def ora_opt(kind, groups, hyper):
        return OO.AdamW(groups, **hyper) if kind == "adamw" else OO.QHM(groups, **hyper)

    torch.set_num_threads(1)
    out = {}
    for name in OC.CASES:
        ref = OC.run_case(ref_opt, name)
        ora = OC.run_case(ora_opt, name)
        for k in range(OC.NSTEPS):
            for i, (a, b) in enumerate(zip(ref[k], ora[k])):
                assert torch.equal(a, b), (name, k, i, (a - b).abs().max())
        for k in (0, OC.NSTEPS - 1):
            for i, a in enumerate(ref[k]):
                out[f"{name}/step{k + 1}/p{i}"] = a.numpy()
        print(name, "oracle == reference (bit-identical) over", OC.NSTEPS, "steps")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "optim_reference.npz"), **out)
    print("wrote tests/golden/optim_reference.npz")


if __name__ == "__main__":
    main()
