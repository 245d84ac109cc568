#!/bin/bash
# Build a named variant of libsmplsim_hip.so for same-box A/B runs (selected at run time with SMPLSIM_HIP_LIB=...).
# usage: tools/build_variant.sh NAME [extra hipcc flags for the stepper translation unit...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OPT=${SS_HIPCC_OPT:--Os -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp}
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OPT "$@" -std=c++17 -fPIC -c smplsim_amd/csrc/smplsim_hip.hip -o build/variants/hip_$NAME.o &
SCOPT=${SS_HIPCC_SC_OPT:--O2 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp}
/opt/rocm/bin/hipcc --offload-arch=gfx950 $SCOPT "$@" -std=c++17 -fPIC -c smplsim_amd/csrc/smplsim_hip_sc.hip -o build/variants/hip_sc_$NAME.o &
# (the imitation unit keeps the shipped flags unless SS_HIPCC_IM_OPT says otherwise: clang 22 crashes in its register allocator on that unit at -O2 / -O3)
IMOPT=${SS_HIPCC_IM_OPT:--Os -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp}
/opt/rocm/bin/hipcc --offload-arch=gfx950 $IMOPT "$@" -std=c++17 -fPIC -c smplsim_amd/csrc/smplsim_hip_im.hip -o build/variants/hip_im_$NAME.o &
XOPT=${SS_HIPCC_X_OPT:--O2 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp}
/opt/rocm/bin/hipcc --offload-arch=gfx950 $XOPT "$@" -std=c++17 -fPIC -c smplsim_amd/csrc/smplsim_hip_x.hip -o build/variants/hip_x_$NAME.o &
wait
[ -f build/variants/motion.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c smplsim_amd/csrc/smplsim_motion.hip -o build/variants/motion.o
[ -f build/variants/mlp.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c smplsim_amd/csrc/smplsim_mlp.hip -o build/variants/mlp.o
mkdir -p smplsim_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/variants/hip_$NAME.o build/variants/hip_sc_$NAME.o build/variants/hip_im_$NAME.o build/variants/hip_x_$NAME.o build/variants/motion.o build/variants/mlp.o -o smplsim_amd/variants/libsmplsim_hip_$NAME.so
echo built smplsim_amd/variants/libsmplsim_hip_$NAME.so
