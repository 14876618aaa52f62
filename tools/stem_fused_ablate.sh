#!/bin/bash
# tools/stem_fused_ablate.sh build "<abl masks>"   (here, no GPU)  -> build_variants/libsi_hip_fusedabl<mask>.so: the product objects with
#                                                   conv_stem_s2c32_f16.hip recompiled under -DSI_FUSED_ABL=<mask> (timing only, wrong results)
# tools/stem_fused_ablate.sh run "<abl masks>"     (GPU box) -> ms of the fused launch per mask (tools/stem_fused_bench.py)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  python -m simpleinfer_amd.build > /dev/null
  mkdir -p build_variants/obj_fused
  for m in $2; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isimpleinfer_amd/csrc/hip -DSI_FUSED_ABL=$m \
        -c simpleinfer_amd/csrc/hip/conv_stem_s2c32_f16.hip -o build_variants/obj_fused/abl_$m.o
      objs=$(ls simpleinfer_amd/build/hip/*.o | grep -v conv_stem_s2c32_f16.hip.o)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_variants/obj_fused/abl_$m.o -o build_variants/libsi_hip_fusedabl$m.so
      echo built build_variants/libsi_hip_fusedabl$m.so ) &
  done
  wait
else
  for m in $2; do
    lib=simpleinfer_amd/libsi_hip.so; [ "$m" != 0 ] && lib=build_variants/libsi_hip_fusedabl$m.so
    echo "=== SI_FUSED_ABL=$m"
    SI_HIP_LIB=$lib python tools/stem_fused_bench.py 2>&1 | grep -E "fused" | tail -1
  done
fi
