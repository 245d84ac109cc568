#!/bin/bash
# Register / scratch / LDS / occupancy of every ss_env_kernel instantiation as hipcc reports them (cross-compile, no GPU needed).
# usage: tools/kernel_resources.sh [extra hipcc flags]     (default flags = smplsim_amd/_lib.py DEFAULT_OPT)
cd "$(dirname "$0")/.."
OPT=${SS_HIPCC_OPT:--Os -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp}
for unit in smplsim_hip smplsim_hip_sc smplsim_hip_im; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OPT "$@" -std=c++17 -fPIC -c smplsim_amd/csrc/$unit.hip -o /tmp/kres_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs:|LDS Size" | \
  sed 's/.*remark: [^ ]* //; s/ \[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - - | grep ss_env_kernel | tr -s ' \t' ' ' | sed -E 's/Name: _ZN[0-9a-zA-Z_]*ss_env_kernelILi([0-9])ELi[0-9]ELi[0-9]ELi[0-9]ELi([0-9]+)ELb([01])ELb([01])E[A-Za-z0-9_]*/K<dofp=\1,maxt=\2,bodyout=\3,shaped=\4>/'
done
rm -f /tmp/kres_$$.o
