#!/bin/bash
# Disassemble one kernel of the built library: tools/dump_isa.sh <mangled-name-substring> [lib]
PAT=$1; LIB=${2:-lib/libtetris_piclim.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $LIB
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $TMP/dev.co | awk -v pat="$PAT" '
  /^[0-9a-f]+ <.*>:$/ { on = (index($0, pat) > 0) }
  on { print }'
rm -rf $TMP
