#!/bin/bash
# tools/hip_variant.sh FILE.hip TAG "DEFINE1 DEFINE2=V ..."  (here, no GPU) -> build_variants/libsi_hip_TAG.so: the product objects with
# ONE translation unit recompiled under the given defines (ablations, ring depths; timing experiments).  Select it on the GPU box
# with SI_HIP_LIB=build_variants/libsi_hip_TAG.so (simpleinfer_amd/_native.py honours it; the product library is never touched).
set -e
cd "$(dirname "$0")/.."
f=$1; tag=$2; defs=""
for d in $3; do defs="$defs -D$d"; done
python -m simpleinfer_amd.build > /dev/null
mkdir -p build_variants/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isimpleinfer_amd/csrc/hip $defs -c simpleinfer_amd/csrc/hip/$f -o build_variants/obj/${f}_$tag.o
objs=$(ls simpleinfer_amd/build/hip/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_variants/obj/${f}_$tag.o -o build_variants/libsi_hip_$tag.so
echo built build_variants/libsi_hip_$tag.so
