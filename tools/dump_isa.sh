#!/bin/bash
# Disassemble one kernel of the built library: tools/dump_isa.sh <mangled-name-substring> [lib]
# (the fat binary holds one offload bundle per translation unit; all of them are searched)
PAT=$1; LIB=${2:-lib/libtetris_piclim.so}
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
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $b.co | awk -v pat="$PAT" '
    /^[0-9a-f]+ <.*>:$/ { on = (index($0, pat) > 0) }
    on { print }'
done
rm -rf $TMP
