#!/usr/bin/env python3
"""Generates tests/golden/genotype_cases.json: inputs + genotype likelihoods of the REAL whatshap.core.GenotypeDPTable
(oracle/_ref/cy, built from /root/reference by oracle/build_cython_ref.py).  Run where /root/reference exists:

    python tests/golden/make_genotype_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from genotype_cases import random_case, reference_likelihoods  # noqa: E402
from helpers import problem_to_json  # noqa: E402
from oracle import build_cython_ref  # noqa: E402


def main():
    ref = build_cython_ref.import_reference()
    cases = []
    specs = [("single", dict(n_variants=10, n_reads=14, max_len=5, max_coverage=7)),
             ("single", dict(n_variants=25, n_reads=60, max_len=8, max_coverage=10, phred=(0, 50))),
             ("trio", dict(n_variants=10, n_reads=12, max_len=4, max_coverage=5)),
             ("trio", dict(n_variants=20, n_reads=40, max_len=6, max_coverage=8, uniform_prior=True)),
             ("quartet", dict(n_variants=8, n_reads=12, max_len=4, max_coverage=5)),
             ("quartet", dict(n_variants=12, n_reads=24, max_len=5, max_coverage=6, uniform_prior=True))]
    for index, (mode, kw) in enumerate(specs):
        p = random_case(31000 + index, mode=mode, **kw)
        gl = reference_likelihoods(p, ref)
        cases.append({"name": f"{mode}_{index}", "problem": problem_to_json(p), "likelihoods": [[[repr(float(x)) for x in col] for col in ind] for ind in gl]})
    with open(os.path.join(HERE, "genotype_cases.json"), "w") as f:
        json.dump({"generated_by": "tests/golden/make_genotype_golden.py", "reference": "whatshap.core.GenotypeDPTable (long double), values as repr(float)",
                   "cases": cases}, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
