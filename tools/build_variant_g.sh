#!/bin/bash
# build_variant_g.sh <name> <extra hipcc flags...>: libgalsynth with synth_group.hip (k_synth_g) built with extra flags, everything
# else the product's objects -> galileo-sdr-sim_amd/variants/libgalsynth_<name>.so (GAL_SYNTH_LIB=..., tools/ab_lib.sh)
set -e
cd "$(dirname "$0")/../galileo-sdr-sim_amd"
name=$1; shift
mkdir -p variants/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS "$@" -c csrc/synth_group.hip -o variants/obj_$name/g.o
make -s csrc/synth_kernels_walk.o csrc/synth_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libgalsynth_$name.so csrc/synth_kernels_walk.o csrc/synth_kernels_f?.o variants/obj_$name/g.o csrc/synth_api.o
ls -la variants/libgalsynth_$name.so
