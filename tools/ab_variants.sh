#!/bin/bash
# tools/ab_variants.sh <bench args...> -- same-box A/B of differently built libsi_hip.so files: every build_variants/libsi_hip_<TAG>.so
# is swapped in turn into the package and `bench.py <args>` run on it; prints value + per-kernel ms.  (GPU box only.)
cp simpleinfer_amd/libsi_hip.so /tmp/libsi_hip_orig.so
for f in build_variants/libsi_hip_*.so; do
  tag=$(basename $f .so); tag=${tag#libsi_hip_}
  cp $f simpleinfer_amd/libsi_hip.so
  python bench.py --no-cpu-baseline --no-aux "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['conv_kernels'].items()})"
done
cp /tmp/libsi_hip_orig.so simpleinfer_amd/libsi_hip.so
