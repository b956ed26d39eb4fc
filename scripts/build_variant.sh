#!/bin/bash
# A/B builds of libpinhip.so with compile-time switches: scripts/build_variant.sh NAME -DFLAG [-DFLAG ...]
# -> pin_slam_amd/_variants/libpinhip_NAME.so; run with PIN_LIBPINHIP=<that path>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
O=$R/pin_slam_amd/_variants/obj_$N; mkdir -p $O
for f in $R/pin_slam_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $f -o $O/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl -o $R/pin_slam_amd/_variants/libpinhip_$N.so $O/*.o
rm -rf $O
ls -la $R/pin_slam_amd/_variants/libpinhip_$N.so
