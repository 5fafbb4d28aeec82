"""No GPU needed: the assembly hipcc produces for gfx950 must not wait for memory right behind a load where round 4 removed such waits.

`x = cond ? p[i] : 0` compiles into a branch around the load that ends in `s_waitcnt vmcnt(0)`: a "batch" of eight loads became eight memory
round trips in a row in geno_slot_combine (24 ms for a trio's 42 GB; 12 ms with unconditional loads from clamped addresses).
scripts/isa_wait_audit.py counts full waits at most three instructions behind a load; this pins the counts of the kernels that were fixed.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def audit(source, pattern):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_wait_audit.py"), source, pattern], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-400:]
    rows = {}
    for line in res.stdout.splitlines()[1:]:
        m = re.match(r"(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+) \|\s+(\d+)\s+(\d+)\s+(\d+)$", line)
        if m:
            rows[m.group(1).strip()] = tuple(int(v) for v in m.groups()[1:])   # loads, waits, full, immediate | prologue loads, full, immediate
    return rows


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
def test_the_genotyping_combine_requests_everything_before_it_waits():
    rows = audit("genotype_slots.hip", "geno_slot_combine")
    assert set(rows) == {"geno_slot_combine<0, 2>", "geno_slot_combine<2, 4>", "geno_slot_combine<4, 4>"}, rows
    for name, (loads, waits, full, immediate, p_loads, p_full, p_immediate) in rows.items():
        assert loads >= 20, (name, loads)          # eight (forward, backward) pairs + the lane table + the block's table pieces
        assert p_loads == loads, (name, "every load belongs to the prologue (before the first barrier)")
        assert immediate <= 1 and p_full <= 2, (name, rows[name], "a load of the combine is waited for right behind its issue again")
