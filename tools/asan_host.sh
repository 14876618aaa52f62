#!/bin/bash
# tools/asan_host.sh -- AddressSanitizer + UBSan build of the HOST library (the GPU build cannot be sanitised on this
# pool) and a run of the loader / ABI tests plus a mutation fuzz of the pnnx loader under it.  CPU only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/si_asan}
mkdir -p "$OUT"
srcs=$(ls "$ROOT"/simpleinfer_amd/csrc/host/*.cpp "$ROOT"/simpleinfer_amd/csrc/host/pnnx/*.cpp "$ROOT"/simpleinfer_amd/csrc/host/layer/*.cpp)
g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" \
    -I"$ROOT/simpleinfer_amd/csrc/host" $srcs -L"$ROOT/simpleinfer_amd" -lsi_hip -Wl,-rpath,"$ROOT/simpleinfer_amd" \
    -o "$OUT/libsimpleinfer_amd.so"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export SI_HOST_LIB="$OUT/libsimpleinfer_amd.so"
cd "$ROOT"
python -m pytest -x -q tests/test_pnnx_loader.py tests/test_abi.py -p no:cacheprovider
python tests/fuzz_loader.py 1500
