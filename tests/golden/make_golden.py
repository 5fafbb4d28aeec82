"""Generates the golden fixtures from the COMPILED REFERENCE (oracle/_ref, built from /root/reference/src).

Run in the build container (where /root/reference exists):   python tests/golden/make_golden.py
Outputs (committed):
  reference_cases.json   the reference's own known-answer unit-test inputs (tests/reference_cases.py) -> outputs
  random_tie_heavy.json  240 seeded tie-heavy random instances (SURVEY.md Appendix A generator)
  synthetic_small.json   small seeded synthetic blocks (single / trio / distrust / step 1 / BLANK-heavy)
Each record holds the flattened problem (the C-ABI views as lists) and the reference's
(cost, index path, transmission vector, partitioning, superreads with qualities), or its error message.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import problem_to_json, table_solution  # noqa: E402
from reference_cases import all_cases  # noqa: E402

import oracle  # noqa: E402
from whatshap_amd.core import problem_from_objects  # noqa: E402
from whatshap_amd.synthetic import random_small_instance, synthetic_block  # noqa: E402


def solve(problem):
    try:
        return {"solution": table_solution(oracle.ReferenceTable(problem)), "error": None}
    except oracle.OracleError as e:
        return {"solution": None, "error": str(e)}


def main():
    oracle.build()
    assert oracle.have_reference(), "oracle/_ref is not built (needs /root/reference)"
    records = []
    for case in all_cases():
        p = problem_from_objects(case.readset, case.recombcost, case.pedigree, case.distrust_genotypes, case.positions)
        rec = {"name": case.name, "problem": problem_to_json(p)}
        rec.update(solve(p))
        assert rec["error"] is None, (case.name, rec["error"])
        if case.expected_cost is not None:
            assert rec["solution"]["cost"] == case.expected_cost, case.name
        records.append(rec)
    json.dump(records, open(os.path.join(HERE, "reference_cases.json"), "w"), separators=(",", ":"))

    rng = random.Random(20250711)
    records = []
    for i in range(240):
        p = random_small_instance(rng)
        rec = {"name": f"random_{i}", "problem": problem_to_json(p)}
        rec.update(solve(p))
        records.append(rec)
    json.dump(records, open(os.path.join(HERE, "random_tie_heavy.json"), "w"), separators=(",", ":"))

    records = []
    for kw in [dict(n_variants=60, coverage=6, seed=2), dict(n_variants=48, coverage=9, seed=4, trio=True),
               dict(n_variants=40, coverage=6, seed=5, trio=True, distrust_genotypes=True),
               dict(n_variants=60, coverage=8, seed=7, distrust_genotypes=True), dict(n_variants=30, coverage=10, seed=13, step=1),
               dict(n_variants=50, coverage=7, seed=21, drop_rate=0.5), dict(n_variants=40, coverage=12, seed=3)]:
        p = synthetic_block(**kw)
        rec = {"name": "synthetic_" + "_".join(f"{k}{v}" for k, v in kw.items()), "problem": problem_to_json(p)}
        rec.update(solve(p))
        records.append(rec)
    json.dump(records, open(os.path.join(HERE, "synthetic_small.json"), "w"), separators=(",", ":"))
    for name in ("reference_cases.json", "random_tie_heavy.json", "synthetic_small.json"):
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
