#!/bin/bash
# tools/detect_ablate.sh build "<abl masks>"   (here, no GPU)  -> build_variants/libsi_hip_detabl<mask>.so: the product objects with
#                                               conv_igemm_f16.hip recompiled under -DSI_DET_ABL=<mask> (timing only, wrong results)
# tools/detect_ablate.sh run "<abl masks>" [batch]   (GPU box) -> ms per Detect level per mask (tools/detect_bench_f16.py)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  python -m simpleinfer_amd.build > /dev/null
  mkdir -p build_variants/obj_detabl
  for m in $2; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isimpleinfer_amd/csrc/hip -DSI_DET_ABL=$m \
        -c simpleinfer_amd/csrc/hip/conv_igemm_f16.hip -o build_variants/obj_detabl/f16_$m.o
      objs=$(ls simpleinfer_amd/build/hip/*.o | grep -v conv_igemm_f16.hip.o)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_variants/obj_detabl/f16_$m.o -o build_variants/libsi_hip_detabl$m.so
      echo built build_variants/libsi_hip_detabl$m.so ) &
  done
  wait
else
  for m in $2; do
    lib=simpleinfer_amd/libsi_hip.so; [ "$m" != 0 ] && lib=build_variants/libsi_hip_detabl$m.so
    echo "=== SI_DET_ABL=$m"
    SI_HIP_LIB=$lib python tools/detect_bench_f16.py $3 2>&1 | grep -E "level|three"
  done
fi
