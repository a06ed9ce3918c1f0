#!/bin/bash
# usage: tools/diag/build_variant.sh <name> <extra hipcc flags...>  -> csrc/libptk_<name>.so (diagnostic builds for timing experiments;
# selected with PTK_FLAVOUR=<name>).  Never part of the product.
set -e
name=$1; shift
cd "$(dirname "$0")/../../project3-cuda-path-tracer-1_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -fno-slp-vectorize"
for f in ptk_kernels dev_scene pathtrace; do /opt/rocm/bin/hipcc $F "$@" -c $f.hip -o /tmp/${f}_$name.o & done
# the second translation unit of ptk_kernels.hip (BSDF-mode k_shade_trace; Makefile: BSDF_TU_FLAGS)
# BSDF_TU_EXTRA="<flags>" replaces that unit's scheduling flags (experiments on the headline's kernel alone)
/opt/rocm/bin/hipcc $F -DPTK_TU_BSDF=1 -Wno-unused-variable ${BSDF_TU_EXTRA:--mllvm -amdgpu-sched-strategy=max-memory-clause} "$@" -c ptk_kernels.hip -o /tmp/ptk_kernels_bsdf_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libptk_$name.so /tmp/ptk_kernels_$name.o /tmp/ptk_kernels_bsdf_$name.o /tmp/dev_scene_$name.o /tmp/pathtrace_$name.o scene.o image.o wide_bvh.o -lz
ls -la libptk_$name.so
