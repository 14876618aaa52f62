#!/bin/bash
# tools/exp_run.sh TAG... -- for each build_variants/libsi_hip_TAG.so: phase stamps on a few shapes, then the whole network
# (selected through SI_HIP_LIB, which simpleinfer_amd/_native.py honours: the product library is never overwritten).  GPU box
# only; same box, back to back.
for tag in "$@"; do
  f=build_variants/libsi_hip_$tag.so
  echo "=================== $tag"
  SI_HIP_LIB=$f python tools/conv_diag.py --shape 32,80,80,128,256,3,2,1 --shape 32,40,40,256,512,3,2,1 --shape 32,320,320,32,64,3,2,1 \
      --shape 32,40,40,256,256,1,1,0 --shape 32,80,80,64,64,1,1,0 --shape 32,20,20,512,512,1,1,0 2>&1 | grep -E "TF/s|cycles per|busy"
  SI_HIP_LIB=$f python bench.py --no-cpu-baseline --no-aux --min-time 1.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('NET $tag', d['value'], d['ms_per_step'], {k.replace('conv_igemm_f32_fast_kernel','fast').replace('conv_',''): (v['ms_per_step'], v['tflops']) for k, v in d['conv_kernels'].items()})"
done
