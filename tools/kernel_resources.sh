#!/bin/bash
# VGPR / SGPR / scratch / LDS of every kernel in the built library (reads the code objects' metadata notes; the fat binary
# holds one offload bundle per translation unit).
LIB=${1:-lib/libtetris_piclim.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB
python3 - $TMP <<'PY'
import sys, os
d = sys.argv[1]
blob = open(os.path.join(d, "fat.bin"), "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
for n, s in enumerate(starts):
    e = starts[n + 1] if n + 1 < len(starts) else len(blob)
    open(os.path.join(d, f"bundle{n}.bin"), "wb").write(blob[s:e])
PY
for b in $TMP/bundle*.bin; do
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$b --output=$b.co --unbundle 2>/dev/null
  [ -s $b.co ] || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $b.co | python3 -c "
import re,sys
txt=sys.stdin.read()
for b in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(k+r':\s+(\S+)', b) or [None,'?'])[1]
    print(f\"{g('.vgpr_count'):>4} vgpr {g('.sgpr_count'):>4} sgpr {g('.private_segment_fixed_size'):>5} scratch {g('.group_segment_fixed_size'):>7} lds  {g('.name')[:110]}\")
"
done
rm -rf $TMP
