#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra -D flags for the fp32 kernel TU>"  -> build/variants/librrt_<name>.so
# A kernel-tuning variant of librrt.so (only device/rrt_f32.hip is recompiled); run it with RRT_LIBRARY=build/variants/librrt_<name>.so
set -e
cd "$(dirname "$0")/../rs_ray_toy_amd/csrc"
out=../../build/variants; mkdir -p $out
make -s librrt.so
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -I../../include -Ihost -Idevice -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $2 -c device/rrt_f32.hip -o $out/rrt_f32_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/librrt_$1.so host/*.o $out/rrt_f32_$1.o device/rrt_f64.o device/rrt_api.o device/rrt_comm.o -lz -lpthread -ldl
echo built $out/librrt_$1.so
