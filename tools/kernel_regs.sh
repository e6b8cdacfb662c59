#!/bin/bash
# usage: tools/kernel_regs.sh <file.hip> [grep-pattern]   - VGPRs / spills / LDS per kernel (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/../qat-vit_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/_regs.o 2>&1 \
 | grep -E "Function Name|    VGPRs:|VGPRs Spill|LDS Size" | sed -E 's/.*remark: +//; s/\[-Rpass.*//' | paste - - - - \
 | sed -E 's/Function Name: //' | c++filt | sed -E 's/void qv:://; s/\(qv::[A-Za-z]*\)//' | grep -E "${2:-.}"
