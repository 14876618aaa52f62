#!/bin/bash
# tools/build_exp.sh TAG "DEFINE1 DEFINE2=..."  -- diagnostic variant of the kernel library: build_variants/libsi_hip_TAG.so
SI_DIAG_TAG=$1 SI_DIAG_DEFINES="$2" python tools/conv_diag.py --build 2>&1 | grep -v "^+" | tail -1
