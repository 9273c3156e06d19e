#!/bin/bash
# Build a variant of librac_hip.so with extra compiler flags, for A/B runs on the GPU box:
#   bash tools/build_variant.sh <name> [-DRAC_EXP_...=1 ...]   ->  robot_aware_control_amd/variants/librac_<name>.so
#   RAC_HIP_LIB=robot_aware_control_amd/variants/librac_<name>.so python tools/bench_gemm.py ...
set -eo pipefail
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/rac_variant_$name; mkdir -p "$obj" "$root/robot_aware_control_amd/variants"
for f in rac_igemm rac_split16 rac_pointwise rac_frame; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$root/include" -Wall -Wno-unused-function \
    -Xclang -target-feature -Xclang -packed-fp32-ops "$@" \
    -c "$root/robot_aware_control_amd/csrc/$f.hip" -o "$obj/$f.o" &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$obj"/*.o -o "$root/robot_aware_control_amd/variants/librac_$name.so"
echo "built variants/librac_$name.so"
