#!/bin/bash
# registers / scratch / LDS of the gfx950 kernels inside a built library:  bash tools/co_regs.sh LIB [name-filter]
# (the code object is cut out of the .hip_fatbin section and its kernel descriptors' metadata notes are read with llvm-readelf)
LIB=$1; FILT=${2:-.}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/co.o 2>/dev/null || \
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hip-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/co.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co.o | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    n=g('name')
    if re.search(r'''$FILT''', n): print('%-60s vgpr %s agpr %s sgpr %s scratch %s lds %s'%(n[:60],g('vgpr_count'),blk.split()[0].strip(':') if False else g('agpr_count') ,g('sgpr_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
"
rm -rf $T
