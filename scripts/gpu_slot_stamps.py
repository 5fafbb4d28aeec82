"""In-kernel cycle stamps of the slot-run kernel (WHAMD_SLOT_STAMPS=1), for the variants of WHAMD_SLOT_SKIP."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WHAMD_SLOT_STAMPS"] = "1"
from whatshap_amd import _native
_native.use_debug_library()   # the timing switches and cycle stamps exist in libwhatshap_amd_debug.so only (csrc/debug_build.h)
from whatshap_amd.synthetic import synthetic_block

p = synthetic_block(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 20, seed=3)
VARIANTS = ((0, ()), (3, ()), (11, ()), (27, ()), (0, (("slot_r", "3"),)), (11, (("slot_r", "3"),)), (0, (("slot_r", "1"),)), (11, (("slot_r", "1"),)))
if len(sys.argv) > 2:
    VARIANTS = tuple(VARIANTS[int(i)] for i in sys.argv[2].split(","))
for skip, options in VARIANTS:
    os.environ["WHAMD_SLOT_SKIP"] = str(skip)
    t = _native.NativeTable(p, solve=False)
    for k, v in options:
        t.set_option(k, v)
    t.solve()
    print(f"-- skip {skip} {options}: forward {t.stats()['forward_ms']:.2f} ms / {t.stats()['forward_launches']} launches", file=sys.stderr, flush=True)
    t.solve()
    t.close()
