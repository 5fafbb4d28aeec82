"""Generates tests/golden/heuristic_cases.json from the COMPILED REFERENCE's PedMecHeuristic (oracle/_ref: src/pedmecheuristic.cpp
+ oracle/ref_driver.cpp).  Run in the build container:   python tests/golden/make_heuristic_golden.py
Each record: the flattened problem, row_limit, and the reference's (score, bipartition, transmission, haplotypes, mutations)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import problem_to_json  # noqa: E402
from heuristic_cases import random_cases, synthetic_cases  # noqa: E402

import oracle  # noqa: E402


def main():
    oracle.build()
    assert oracle.have_reference(), "oracle/_ref is not built (needs /root/reference)"
    records = []
    for name, problem, row_limit in random_cases(20250925, 25) + synthetic_cases()[:4]:
        want = oracle.heuristic_tuple(oracle.ReferenceHeuristic(problem, row_limit=row_limit))
        records.append({"name": name, "row_limit": row_limit, "problem": problem_to_json(problem), "solution": want})
    path = os.path.join(HERE, "heuristic_cases.json")
    json.dump(records, open(path, "w"), separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes,", len(records), "records")


if __name__ == "__main__":
    main()
