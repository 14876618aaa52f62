#!/bin/bash
# tools/f16_ablate.sh build "<abl masks>"   (here, no GPU)  -> build_variants/libsi_hip_f16abl<mask>.so: the product objects with
#                                             conv_igemm_f16.hip recompiled under -DSI_F16_ABL=<mask> (timing only, wrong results)
# tools/f16_ablate.sh run "<abl masks>" <variant> [conv_bench args]   (GPU box) -> us per launch per mask on the given tile variant
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  python -m simpleinfer_amd.build > /dev/null
  mkdir -p build_variants/obj_f16abl
  for m in $2; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isimpleinfer_amd/csrc/hip -DSI_F16_ABL=$m \
        -c simpleinfer_amd/csrc/hip/conv_igemm_f16.hip -o build_variants/obj_f16abl/f16_$m.o
      objs=$(ls simpleinfer_amd/build/hip/*.o | grep -v conv_igemm_f16.hip.o)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_variants/obj_f16abl/f16_$m.o -o build_variants/libsi_hip_f16abl$m.so
      echo built build_variants/libsi_hip_f16abl$m.so ) &
  done
  wait
else
  masks=$2; v=$3; shift; shift; shift
  for m in $masks; do
    lib=simpleinfer_amd/libsi_hip.so; [ "$m" != 0 ] && lib=build_variants/libsi_hip_f16abl$m.so
    echo "=== SI_F16_ABL=$m (variant $v)"
    SI_HIP_LIB=$lib SI_CONV_F16_VARIANT=$v python tools/conv_bench.py --f16 --min-ms 30 "$@" 2>&1 | grep -E "k[0-9]s[0-9]"
  done
fi
