"""GPU (-m gpu): untrusted trios as ONE sequence of launches -- `pedslot_group<2, PSLOT_FACT>` (whamd_dptable_enqueue_many on tables whose cost
lines are factorised, slots.h) next to other kernel variants in the same super-steps, every table against the oracle.

(Its own file, after the others in collection order: the variant was added after the round's last full run on the device.)
"""
import pytest

import oracle
from helpers import first_difference, table_solution
from whatshap_amd import _native
from whatshap_amd.synthetic import synthetic_block

pytestmark = pytest.mark.gpu


def test_untrusted_trios_share_their_launches_on_factorised_lines():
    """Seven trio tables whose genotypes are not trusted (factorised cost lines, slots.h PSLOT_FACT: `pedslot_group<2, PSLOT_FACT>`), widths from
    64-thread workgroups to full ones, next to a trusted trio and a single individual (their own kernel variants in the same super-steps):
    batched == the oracle for every table."""
    cases = [synthetic_block(n_variants=n, coverage=cov, seed=400 + i, trio=True, distrust_genotypes=True, mixed_genotypes=i % 2 == 1)
             for i, (cov, n) in enumerate([(5, 200), (7, 300), (9, 500), (10, 700), (11, 900), (12, 1200), (13, 800)])]
    cases += [synthetic_block(n_variants=400, coverage=10, seed=410, trio=True), synthetic_block(n_variants=600, coverage=12, seed=411)]
    assert all(_native.plan_summary(p)["n_fact_runs"] > 0 for p in cases[:7])
    tables = [_native.NativeTable(p, solve=False) for p in cases]
    _native.enqueue_many(tables)
    _native.wait_many(tables)
    assert max(t.stats()["group_tables"] for t in tables) >= 7, "the untrusted trios did not share their launches"
    for i, (p, t) in enumerate(zip(cases, tables)):
        want, got = table_solution(oracle.OracleTable(p)), table_solution(t)
        assert got == want, (i, first_difference(want, got))
        t.close()
