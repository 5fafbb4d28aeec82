"""GPU run (-m gpu) of the read-selection row: host code by nature, but part of the product library the
driver loads on the GPU box -- so its comparisons with the built reference module (whatshap.readselect, oracle/_ref/cy)
run there too: same interpreter and libstdc++ as the one the tie order was pinned on (csrc/readselect.cpp:40-176).
The tests themselves live in test_readselect.py (they also run in the CPU selection)."""
import pytest

from test_readselect import (test_accepts_a_reference_readset, test_many_reads_beyond_the_x4_growth_of_sets,  # noqa: F401
                             test_preferred_sources_first, test_same_selection_as_the_reference_module,
                             test_selection_respects_the_coverage_bound)

pytestmark = pytest.mark.gpu
