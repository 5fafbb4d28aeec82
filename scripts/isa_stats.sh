#!/bin/bash
# Register / LDS / scratch use of the kernels of one device source (cross-compiled, no GPU needed):  scripts/isa_stats.sh dp_device.hip [name filter]
set -e
src=${1:-dp_device.hip}; filt=${2:-.}
out=/tmp/whamd_isa; mkdir -p $out
cd "$(dirname "$0")/../whatshap_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$src" -o $out/${src%.hip}.s 2>/dev/null
python3 - "$out/${src%.hip}.s" "$filt" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
md = txt[txt.index('amdhsa.kernels'):]
for e in md.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', e).group(1)
    if not re.search(sys.argv[2], name):
        continue
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', e).group(1)
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>4} spills {g('vgpr_spill_count')}")
PY
