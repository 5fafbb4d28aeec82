import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whatshap_amd import _native
_native.use_debug_library()   # the timing switches and cycle stamps exist in libwhatshap_amd_debug.so only (csrc/debug_build.h)
from whatshap_amd.synthetic import random_small_instance
from oracle import OracleTable
rng = random.Random(11)
for it in range(400):
    p = random_small_instance(rng)
    if it == 8:
        break
o = OracleTable(p)
print("oracle", o.index_path()[0].tolist(), o.optimal_score())
os.environ["WHAMD_DEBUG_PLAN"] = "1"
t = _native.NativeTable(p, path="resident")
print("native", t.index_path()[0].tolist(), t.optimal_score())
print("reads", p.read_ptr.tolist(), p.var_position.tolist(), p.positions.tolist())
