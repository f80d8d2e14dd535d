#!/bin/bash
# Diagnostic variants of the pair kernel only: tools/variant_fast4.sh <name> <extra hipcc flags...>
# compiles wn_k_fast4.hip (and wn_capi.hip, which prints the stamps) with the flags and links them with the standard
# objects of walnuts_amd/csrc/build -> walnuts_amd/libwalnuts_amd_<name>.so (select it with WN_LIB=<name>)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../walnuts_amd/csrc"
mkdir -p build_$NAME
FLAGS="-O3 -std=c++17 -fPIC -fno-strict-aliasing --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value $*"
/opt/rocm/bin/hipcc $FLAGS -c wn_k_fast4.hip -o build_$NAME/wn_k_fast4.o &
case " $* " in *WN_STAMPS*) /opt/rocm/bin/hipcc $FLAGS -c wn_capi.hip -o build_$NAME/wn_capi.o & ;; *) cp build/wn_capi.o build_$NAME/wn_capi.o ;; esac
wait
OTHERS=$(ls build/*.o | grep -v -e wn_k_fast4.o -e wn_capi.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwalnuts_amd_$NAME.so build_$NAME/wn_k_fast4.o build_$NAME/wn_capi.o $OTHERS
echo built ../libwalnuts_amd_$NAME.so
