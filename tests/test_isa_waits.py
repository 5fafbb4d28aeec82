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


# ---- round 5: the X runs (kernels_slots.h, slot_runx) -- the column trip as it is MEANT to be issued, pinned on the cross-compiled assembly
def _device_asm():
    """dp_device.hip compiled to gfx950 assembly (once per test session: ~1 min)."""
    out = "/tmp/whamd_isa"
    os.makedirs(out, exist_ok=True)
    asm = os.path.join(out, "dp_device_xrun.s")
    src_dir = os.path.join(ROOT, "whatshap_amd", "csrc")
    newest = max(os.path.getmtime(os.path.join(src_dir, n)) for n in os.listdir(src_dir) if n.endswith((".h", ".hip")))
    if not os.path.exists(asm) or os.path.getmtime(asm) < newest:
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "dp_device.hip", "-o", asm], cwd=src_dir, check=True,
                       stderr=subprocess.DEVNULL, timeout=900)
    return open(asm).read()


def _kernel(text, mangled_fragment):
    m = re.search(r"^(_Z\w*" + mangled_fragment + r"\w*):", text, re.M)
    assert m, mangled_fragment
    body = text[m.end():text.index(".Lfunc_end", m.end())]
    meta = text[text.index("amdhsa.kernels"):]
    entry = next(e for e in meta.split("- .agpr_count")[1:] if m.group(1) in e)
    num = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", entry).group(1))
    return body, {k: num(k) for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count")}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
def test_the_x_run_column_trip_has_no_wait_and_no_memory_operation_on_the_plain_path():
    """"a plain column takes 230 cycles for 12 instructions ... the wait structure has not been written by hand".  In an X run a plain
    column is four v_sad_u32 with a scalar second operand, a scalar test and a branch -- nothing else, no wait, no LDS read, no address arithmetic; the
    waits of a trip stand at its head (before the next trip's requests go out) and inside the hand-written ending block."""
    text = _device_asm()
    for frag, spills_ok in (("9slot_runxILi2ELi24ELb0ELb0E", 0), ("9slot_runxILi2ELi32ELb0ELb0E", 0), ("11slot_groupxILi2ELb0E", 0)):
        body, meta = _kernel(text, frag)
        assert meta["vgpr_count"] <= 64 and meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0, (frag, meta)
        assert meta["sgpr_spill_count"] <= spills_ok, (frag, meta)   # (a spilled scalar is a v_readlane per use: round 5's first group kernel had 500 in its loop)
        lines = [ln.strip() for ln in body.split("\n")]
        lines = [ln for ln in lines if ln and not ln.startswith(";") and not ln.startswith(".") or ln.startswith(".LBB") or ln.startswith(".Lx")]
        # the column loop: from the first v_sad_u32 with an SGPR operand to the last
        sads = [i for i, ln in enumerate(lines) if re.match(r"v_sad_u32 v\d+, v\d+, s\d+, v\d+", ln)]
        assert len(sads) == 32, (frag, len(sads))                 # two trips of four columns, four cells each
        columns = [sads[i:i + 4] for i in range(0, 32, 4)]
        for col in columns:
            # the four cells back to back, at most two scalar instructions of the control test between them (the compiler interleaves them)
            between = lines[col[0]:col[3] + 1]
            assert sum(1 for ln in between if not ln.startswith("v_sad_u32")) <= 2, (frag, between)
            # ... then the test "does a read end here": at most three scalar instructions up to the branch, none of them a wait or a memory operation
            tail = []
            for ln in lines[col[3] + 1:]:
                tail.append(ln)
                if ln.startswith("s_cbranch"):
                    break
            assert len(tail) <= 4 and not any(re.match(r"s_waitcnt|ds_|global_|s_load|v_readlane|v_readfirstlane", ln) for ln in tail), (frag, tail)
        loop = lines[sads[0]:sads[-1] + 1]
        # the hand-written ending block: every one of its eight copies (one per column of the two trips) requests the partner cells before it prepares the masks
        starts = [i for i, ln in enumerate(loop) if ln.startswith("s_bfe_u32") and loop[i + 1].startswith("s_cmp_lt_u32") and "8" in loop[i + 1]]
        assert len(starts) >= 8, (frag, len(starts))
        assert sum(1 for ln in loop if ln.startswith("ds_bpermute_b32 v6")) == 4 * len(starts)
        # outside those blocks the loop's waits are the two trip heads (scalar-cache loads return out of order: the wait stands BEFORE the next requests)
        # and the rare second ending read of a column; none of them is a vector-memory wait
        assert not any(re.search(r"vmcnt\(0\)", ln) for ln in loop) or frag.startswith("11"), (frag, "a full vector-memory wait inside the column loop")
        assert sum(1 for ln in loop if ln.startswith("v_readfirstlane")) == 0, (frag, "VALU -> SGPR copies inside the column loop")
    # eight cells per thread (wide tables that share their launches): the same budget of registers -- four workgroups per CU
    body, meta = _kernel(text, "11slot_groupxILi3ELb0E")
    assert meta["vgpr_count"] <= 64 and meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0 and meta["sgpr_spill_count"] == 0, meta
    assert len(re.findall(r"v_sad_u32 v\d+, v\d+, s\d+, v\d+", body)) == 32      # two trips of two columns, eight cells each


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
def test_the_pedigree_runs_min_plus_step_on_packed_keys_is_an_add_with_dpp_and_a_min_per_stage():
    """Round 5 (kernels_pedslots.h): with (value, j) packed into one key a stage of the butterfly over the previous transmission value is the partner's key +
    the scaled recombination cost -- the DPP move folded into the add -- and a v_min; the staged comparison it replaces (kept for tables whose values do
    not leave the room) needs two DPP moves, two saturating adds, a compare and two selects."""
    text = _device_asm()
    packed, meta = _kernel(text, "11pedslot_runILi2ELi2ELb0ELb1E")
    staged, _ = _kernel(text, "11pedslot_runILi2ELi2ELb0ELb0E")
    assert meta["private_segment_fixed_size"] == 0 and meta["vgpr_spill_count"] == 0 and meta["sgpr_spill_count"] == 0, meta
    folded = len(re.findall(r"v_add_u32_dpp", packed))
    assert folded == 8, folded                                    # four columns per trip, two stages each (T = 4)
    assert len(re.findall(r"v_mov_b32_dpp", packed)) == 0         # no DPP move left on its own
    assert len(re.findall(r"v_add_u32_dpp", staged)) == 0 and len(re.findall(r"v_mov_b32_dpp", staged)) == 16
    # a quartet: four stages; bit 3 is ONE row rotation (row_ror:8), bit 2 a half-mirror move + a folded quad permute
    quartet, _ = _kernel(text, "11pedslot_runILi4ELi2ELb0ELb1E")
    assert len(re.findall(r"row_ror:8", quartet)) == 4 and len(re.findall(r"v_add_u32_dpp", quartet)) == 16, (len(re.findall(r"row_ror:8", quartet)), len(re.findall(r"v_add_u32_dpp", quartet)))
