#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"  -> dxrexperiments_amd/lib/variants/lib<name>.so
set -e
N=$1; shift
mkdir -p build_$N dxrexperiments_amd/lib/variants
for f in rt_api.hip rt_bvh_build.hip rt_bvh_ploc.hip rt_bvh_wide.hip rt_trace.hip rt_pipeline.hip rt_pipeline_render.hip rt_pipeline_host.hip rt_denoise.hip rt_dist.hip rt_obj.cpp rt_fbx.cpp rt_host.cpp rt_dds.cpp rt_image.cpp; do
  if [ build_$N/$f.o -nt dxrexperiments_amd/csrc/$f ] && [ "$f" != rt_trace.hip ] && [ "$f" != rt_pipeline.hip ] && [ "$f" != rt_pipeline_render.hip ] && [ "$f" != rt_pipeline_host.hip ]; then continue; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Iinclude $* -x hip -c dxrexperiments_amd/csrc/$f -o build_$N/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o dxrexperiments_amd/lib/variants/lib$N.so build_$N/*.o -ldl -lz
echo built dxrexperiments_amd/lib/variants/lib$N.so
