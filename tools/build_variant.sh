#!/bin/bash
# Builds a kernel variant of liblrltrace.so into gpurun_out/variants/<name>.so (experiments; see tools/sweep notes in DESIGN.md)
# usage: tools/build_variant.sh name "-DLRL_MIN_WAVES_PER_SIMD=4 ..."
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
b=/tmp/lrl_variant_$name; mkdir -p $b $root/raytracing_amd/variants
cd $root/raytracing_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra"
hipcc $F -ffp-contract=off -c kernels_parity.hip -o $b/kp.o &
hipcc $F -ffp-contract=fast -fno-slp-vectorize -c kernels_fast.hip -o $b/kf.o -Rpass-analysis=kernel-resource-usage 2> $b/res.txt &
hipcc $F -c lrl_context.hip -o $b/t1.o &
hipcc $F -c lrl_scene.hip -o $b/t2.o &
hipcc $F -c lrl_schedule.hip -o $b/t3.o &
hipcc $F -c lrl_build.hip -o $b/b.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/raytracing_amd/variants/$name.so $b/kp.o $b/kf.o $b/t1.o $b/t2.o $b/t3.o $b/b.o
grep -A12 "renderKernelILb0ELb0" $b/res.txt | grep -E "VGPRs:|Scratch|Occupancy" | tr '\n' ' '; echo " <- $name"
