#!/bin/bash
# tools/l2_probe.sh ["hot block ticks spin" ...] (GPU box): WRITE_SIZE of tools/l2_writeback_probe.bin for a few hot-region sizes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/l2probe; rm -rf $O; mkdir -p $O
[ $# -gt 0 ] || set -- "1024 99328 2000 10" "8192 99328 2000 10" "28672 99328 2000 10" "28672 28672 2000 10" "3584 99328 4000 10"
for cfg in "$@"; do
  n=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$n -- $R/tools/l2_writeback_probe.bin $cfg > $O/$n.log 2>&1
  python3 - $O/$n "$cfg" <<'PY'
import csv,glob,sys
tot=0
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'probe' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE': tot+=float(r['Counter_Value'])
print(sys.argv[2], 'WRITE_SIZE', tot*1024/1e9, 'GB', open(sys.argv[1]+'.log').read().strip().splitlines()[-1])
PY
done
