#!/bin/bash
# register / scratch / LDS figures of the kernels in a built library: tools/debug/regs.sh [lib] [name filter]
lib=${1:-/root/repo/baryonyx_amd/csrc/libbxh.so}; pat=${2:-.}
d=$(mktemp -d); cd $d
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=fat.bin $lib
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=gfx950.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes gfx950.co | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.sgpr_spill_count|\.vgpr_spill_count|private_segment_fixed_size" | paste - - - - - - | sed 's/ \+/ /g' | grep -E "$pat" | c++filt | cut -c1-260
rm -rf $d
