#!/bin/bash
# VGPR / SGPR / scratch / LDS of every kernel in the built library (reads the code object's metadata notes).
LIB=${1:-lib/libtetris_piclim.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$LIB --output=$TMP/dev.co --unbundle 2>/dev/null \
  || /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB
if [ ! -s $TMP/dev.co ]; then
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle
fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co | python3 -c "
import re,sys
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    pass
blocks=txt.split('- .agpr_count')
for b in blocks[1:]:
    g=lambda k: (re.search(k+r':\s+(\S+)', b) or [None,'?'])[1]
    print(f\"{g('.vgpr_count'):>4} vgpr {g('.sgpr_count'):>4} sgpr {g('.private_segment_fixed_size'):>5} scratch {g('.group_segment_fixed_size'):>7} lds  {g('.name')[:110]}\")
"
rm -rf $TMP
