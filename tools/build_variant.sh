#!/bin/bash
# Build a variant of librac_hip.so with extra compiler flags, for A/B runs on the GPU box:
#   bash tools/build_variant.sh <name> [-DRAC_EXP_...=1 ...]   ->  robot_aware_control_amd/variants/librac_<name>.so
#   RAC_HIP_LIB=robot_aware_control_amd/variants/librac_<name>.so python tools/bench_gemm.py ...
# RAC_PACKED=1 keeps the compiler's packed fp32 forms (v_pk_fma_f32 ...), which the shipped library is built without.
set -eo pipefail
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
nopk="-Xclang -target-feature -Xclang -packed-fp32-ops"; [ "$RAC_PACKED" = "1" ] && nopk=""
obj=/tmp/rac_variant_$name; mkdir -p "$obj" "$root/robot_aware_control_amd/variants"
# RAC_PACKED_FILES="rac_frame rac_split16": packed fp32 ops in those translation units only (bisecting the miscompute)
for f in rac_igemm rac_split16 rac_pointwise rac_frame; do
  fl=$nopk; case " $RAC_PACKED_FILES " in *" $f "*) fl="";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$root/include" -Wall -Wno-unused-function \
    $fl "$@" \
    -c "$root/robot_aware_control_amd/csrc/$f.hip" -o "$obj/$f.o" &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$obj"/*.o -o "$root/robot_aware_control_amd/variants/librac_$name.so"
echo "built variants/librac_$name.so"
