#!/bin/bash
# tools/build_prev.sh [commit]: the library of a commit (default HEAD) as tools/librmhmc_hip_prev.so, for tools/ab.sh
C=${1:-HEAD}
rm -rf /tmp/oldsrc && mkdir -p /tmp/oldsrc && git archive $C riemannhamiltonianmontecarlo_amd/csrc include | tar -x -C /tmp/oldsrc &&
(cd /tmp/oldsrc/riemannhamiltonianmontecarlo_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -o $OLDPWD/tools/librmhmc_hip_prev.so rmhmc_hip.hip)
