#!/bin/bash
# Code-object metadata of one translation unit's kernels (registers, spills, LDS, scratch), from the compiler's own
# assembly: tools/kernel_meta.sh wn_k_fast4 [extra hipcc flags]  ->  /tmp/<unit>.s and a table on stdout.
# Occupancy reasoning cites THESE numbers (arch + accumulation VGPRs, spill counts), not the trace's VGPR_Count column.
set -e
U=$1; shift || true
cd "$(dirname "$0")/../walnuts_amd/csrc"
EXTRA=""; [ $U = wn_k_gemm ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-strict-aliasing --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value $EXTRA "$@" \
  --cuda-device-only -S -o /tmp/$U.s $U.hip
python3 - /tmp/$U.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size: *\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\." + k + r": *(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print(f"{name[:70]:70s} vgpr={g('vgpr_count')} agpr={g('agpr_count')} sgpr={g('sgpr_count')} vgpr_spill={g('vgpr_spill_count')} "
          f"sgpr_spill={g('sgpr_spill_count')} lds={g('group_segment_fixed_size')} scratch={g('private_segment_fixed_size')}")
PY
