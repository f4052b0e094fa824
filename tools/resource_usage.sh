#!/bin/bash
# Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.
cd "$(dirname "$0")/../spmv_acc_amd/csrc"
for f in k_vector_row k_rowblock k_flat k_plus k_segment k_slab k_legacy k_guard k_col16 k_hint k_analyze; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -c $f.hip -o /tmp/$f.ru.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" |
    sed -E 's/.*remark: //; s/ \[-Rpass.*//' | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' |
    sed -E 's/Function Name: _ZN8spmv_acc12_GLOBAL__N_1[0-9]+//; s/EEEv.*\| TotalSGPRs/ | SGPRs/; s/EPK.*\| TotalSGPRs/ | SGPRs/; s/Eid.*\| TotalSGPRs/ | SGPRs/; s/Eii.*\| TotalSGPRs/ | SGPRs/'
done
