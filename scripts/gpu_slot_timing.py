"""Forward / backtrace timings of the slot-run path with parts of the kernel switched off (WHAMD_SLOT_SKIP: results
invalid, timings only), next to the LDS-resident path.  Usage: python scripts/gpu_slot_timing.py [variants] [coverage]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()   # the timing switches and cycle stamps exist in libwhatshap_amd_debug.so only (csrc/debug_build.h)
from whatshap_amd.synthetic import synthetic_block

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 20
p = synthetic_block(n, cov, seed=3)


def run(label, path=None, skip=None, options=()):
    if skip is None:
        os.environ.pop("WHAMD_SLOT_SKIP", None)
    else:
        os.environ["WHAMD_SLOT_SKIP"] = str(skip)
    t = _native.NativeTable(p, solve=False, path=path)
    for k, v in options:
        t.set_option(k, v)
    best = None
    for _ in range(3):
        t.solve()
        s = t.stats()
        if best is None or s["forward_ms"] < best["forward_ms"]:
            best = s
    runs = best["forward_launches"]
    print(f"{label:44s} forward {best['forward_ms']:8.2f} ms  backtrace {best['backtrace_ms']:7.2f} ms  launches {runs:6d}  "
          f"{best['forward_ms'] * 1e3 / max(runs, 1):6.2f} us/launch  {n / (best['forward_ms'] + best['backtrace_ms']) / 1e3:7.3f} M col/s (device)", flush=True)
    t.close()


run("slots")
run("slots, no exit stores (1)", skip=1)
run("slots, no records (2)", skip=2)
run("slots, no stores at all (3)", skip=3)
run("slots, one column per run (4)", skip=4)
run("slots, one column, no stores (7)", skip=7)
run("slots, no ending reads (8)", skip=8)
run("slots, no ending reads, no stores (11)", skip=11)
run("slots, no endings, no cost, no stores (27)", skip=27)
run("slots, no cost, no stores (19)", skip=19)
for r, l in (("3", "11"),):
    run(f"slots, slot_r={r} slot_l={l}", options=(("slot_r", r), ("slot_l", l)))
    run(f"slots, slot_r={r} no stores (3)", skip=3, options=(("slot_r", r), ("slot_l", l)))
    run(f"slots, slot_r={r} no endings no stores (11)", skip=11, options=(("slot_r", r), ("slot_l", l)))
    run(f"slots, slot_r={r} no endings/cost/stores (27)", skip=27, options=(("slot_r", r), ("slot_l", l)))
run("slots, symmetry off", options=(("symmetry", "0"),))
run("resident (LDS runs)", path="resident")
