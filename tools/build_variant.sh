#!/bin/bash
# Build a diagnostic variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>
# -> walnuts_amd/libwalnuts_amd_<name>.so (select it with WN_LIB=<name>)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../walnuts_amd/csrc"
mkdir -p build_$NAME
FLAGS="-O3 -std=c++17 -fPIC -fno-strict-aliasing --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value $*"
BASE="-O3 -std=c++17 -fPIC -fno-strict-aliasing --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value"
# PLAIN="wn_k_split wn_k_wave" compiles those translation units without the extra flags
for f in wn_capi wn_k_lane wn_k_coop wn_k_fast wn_k_fast4 wn_k_wave wn_k_wavelds wn_k_orbit wn_k_orbit2 wn_k_split wn_k_gemm wn_k_wstep; do
  FL="$FLAGS"
  case " ${PLAIN:-} " in *" $f "*) FL="$BASE";; esac
  [ $f = wn_k_gemm ] && FL="$FL -mllvm -amdgpu-mfma-vgpr-form"
  ( /opt/rocm/bin/hipcc $FL -c $f.hip -o build_$NAME/$f.o ) &
done
( /opt/rocm/bin/hipcc $FLAGS -DWN4_WARM_TU=1 -c wn_k_fast4.hip -o build_$NAME/wn_k_fast4w.o ) &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libwalnuts_amd_$NAME.so build_$NAME/*.o
echo built ../libwalnuts_amd_$NAME.so
