#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: libgalsynth built with extra flags for the k_synth families (A/B experiments;
# e.g. tools/build_variant.sh w4 -DSYN_WAVES=4) -> galileo-sdr-sim_amd/variants/libgalsynth_<name>.so, used through
# GAL_SYNTH_LIB=... (synth.load_library); the walker TU and the host code are the product's objects.
set -e
cd "$(dirname "$0")/../galileo-sdr-sim_amd"
name=$1; shift
mkdir -p variants/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-value"
for k in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc $FLAGS -DGAL_TU=$k "$@" -c csrc/synth_kernels.hip -o variants/obj_$name/f$k.o &
done
wait
make -s csrc/synth_kernels_walk.o csrc/synth_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libgalsynth_$name.so csrc/synth_kernels_walk.o variants/obj_$name/f*.o csrc/synth_api.o
ls -la variants/libgalsynth_$name.so
