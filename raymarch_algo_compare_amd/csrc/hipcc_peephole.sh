#!/bin/bash
# hipcc -c with the ISA peephole (rm_peephole.py) between the compiler and the assembler:
#   hipcc_peephole.sh <out.o> <src.hip> <flags...>
# device: hipcc -S -> rm_peephole.py -> assemble -> lld -> offload bundle; host: hipcc --cuda-host-only with that bundle.
# (the same steps `hipcc -c` runs internally: hipcc -### -c shows them)
set -euo pipefail
out="$1"; src="$2"; shift 2
here="$(cd "$(dirname "$0")" && pwd)"
llvm="${ROCM_LLVM:-/opt/rocm/lib/llvm/bin}"
hipcc="${HIPCC:-hipcc}"
base="${out%.o}"
"$hipcc" "$@" -Wno-unused-command-line-argument --cuda-device-only -S "$src" -o "$base.raw.s"
python3 "$here/rm_peephole.py" "$base.raw.s" "$base.s"
"$llvm/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$base.s" -o "$base.dev.o"
"$llvm/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$base.hsaco" "$base.dev.o"
"$llvm/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input="$base.hsaco" -output="$base.hipfb"
"$hipcc" "$@" --cuda-host-only -c "$src" -o "$out" -Xclang -fcuda-include-gpubinary -Xclang "$base.hipfb"
rm -f "$base.raw.s" "$base.dev.o" "$base.hsaco" "$base.hipfb"
[ -n "${KEEP_ASM:-}" ] || rm -f "$base.s"      # KEEP_ASM=1 keeps the rewritten assembly next to the object
