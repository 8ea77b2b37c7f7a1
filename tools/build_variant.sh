#!/bin/bash
# Builds a kernel variant of the library for A/B timing:  tools/build_variant.sh NAME -DVO_HP_U=8 ...
# -> gpurun_variants/libvo_amd_NAME.so ; select it at run time with VO_AMD_LIB=<path>.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=${VO_SRC:-$root/visualodometry_amd/csrc}
out=$root/gpurun_variants
mkdir -p $out/obj_$name
for f in vo_abi fast sgbm lk pnp pnp_solve cloud; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $src/$f.hip -o $out/obj_$name/$f.o &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $src/pnp_host.cpp -o $out/obj_$name/pnp_host.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libvo_amd_$name.so $out/obj_$name/*.o -lpthread
rm -rf $out/obj_$name
echo built $out/libvo_amd_$name.so
