#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> build/ab/libmvrt_<name>.so (A/B builds; load with MVRT_LIB=...)
# built with -DMVRT_EXPERIMENT: the tuning constants of the product (mvrtKnob) can then be overridden from the environment
name=$1; shift
d=build/ab/$name; mkdir -p $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-gpu-rdc -Wno-unused-result -DMVRT_EXPERIMENT $@"
for s in api kernels_rt kernels_setup svo_build; do
  if [ $s = kernels_rt ] || [ ! -f $d/$s.o ]; then /opt/rocm/bin/hipcc $F -c massivevoxelraytracing_amd/csrc/$s.hip -o $d/$s.o & fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libmvrt_$name.so $d/api.o $d/kernels_rt.o $d/kernels_setup.o $d/svo_build.o && echo built build/ab/libmvrt_$name.so
