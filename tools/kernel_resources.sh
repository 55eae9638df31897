#!/bin/bash
# Registers, spills, scratch and occupancy of every kernel of libprhf.so as the backend reports them
# (-Rpass-analysis=kernel-resource-usage); runs without a GPU.  Usage: tools/kernel_resources.sh [extra hipcc flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/pyrayhf_amd/csrc --offload-arch=gfx950 -ffp-contract=off \
    -mllvm -disable-machine-licm "$@" -Rpass-analysis=kernel-resource-usage -c $ROOT/pyrayhf_amd/csrc/prhf_kernels.hip \
    -o $TMP/k.o 2> $TMP/res.txt
grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy|SGPRs:" $TMP/res.txt | sed 's/.*remark: [^ ]* //' | paste - - - - - - - |
    sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g;s/ \+/ /g' | c++filt | cut -c1-260
rm -rf $TMP
