#!/usr/bin/env python3
"""Measured accuracy of the two GEMM modes on adversarial dynamic range (the cases of tests/test_split_range_gpu.py) -> JSON:
relative errors against fp64, split-fp16 engine beside the exact-fp32 matrix-core kernels.  Usage: python tools/split_range_report.py out.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_split_range_gpu as TR  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {"note": "relative errors against an fp64 evaluation; *_row = max over rows of max|err| / max|ref| of the row; envelope <= 1 = "
                   "tests.test_split_range_gpu.row_envelope", "first_block": {}, "attnpool": {}, "full_step_T0.001": {}}
    for kind in TR.KINDS:
        out["first_block"][kind] = TR.case_block1(dev, kind)
    out["first_block"]["zero_bags_no_bias"] = TR.case_block1(dev, "zero_bags", with_bias=False)
    for peak in (1.0, 40.0, 120.0):
        out["attnpool"]["wc_x%g" % peak] = TR.case_attnpool(dev, peak)
    for kind in ("uniform", "outlier_patch_2^20", "student_t2", "zero_bags"):
        out["full_step_T0.001"][kind] = TR.case_full_step(dev, kind)
    out["full_step_T0.01_got"] = {"outlier_patch_2^20": TR.case_full_step(dev, "outlier_patch_2^20", use_got=True, T_=0.01)}
    # the opt-in two-term backward (functional.set_gradient_terms(2)): same cases, split mode only
    from madeleine_amd import functional as MF
    MF.set_gradient_terms(2)
    try:
        out["full_step_T0.001_grad_terms2"] = {k: TR.case_full_step(dev, k)["split"]
                                               for k in ("uniform", "outlier_patch_2^20", "student_t2", "zero_bags")}
        out["first_block_grad_terms2"] = {k: TR.case_block1(dev, k) for k in ("uniform", "outlier_patch_2^20")}
    finally:
        MF.set_gradient_terms(3)
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:3000])


if __name__ == "__main__":
    main()
