#!/bin/bash
# kernel_resources.sh FILE.hip [extra hipcc flags] -- registers, LDS and scratch of every kernel of one translation unit
# (device-only compile to assembly with the library's flags; reads the .amdhsa metadata)
set -e
cd "$(dirname "$0")/../cae-lo_amd/csrc"
f=$1; shift
out=/tmp/isa/$(basename "$f" .hip).s
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Xclang -target-feature -Xclang -packed-fp32-ops \
    -DCAELO_BUILD_WORD=256 $([ "$f" = encoder.hip ] && echo "-mllvm -amdgpu-atomic-optimizer-strategy=None") "$@" --cuda-device-only -S -o "$out" "$f" 2>&1 | grep -v "recognized feature\|hip-link" || true
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    b = m.group(0)
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
    print("%-70s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s" % (g("name")[:70], g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
PY
