cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_funnel
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/p1 -- python3 bench.py --workload funnel --no-extras --no-cpu-baseline --steps 1 --warmup 0 > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2 -- python3 bench.py --workload funnel --no-extras --no-cpu-baseline --steps 1 --warmup 0 > $O/p2.log 2>&1
python3 - <<'PY'
import csv,glob,os
acc={}
for f in glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_funnel/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wn_wave' in r['Kernel_Name']:
            k=(r['Counter_Name'],r['Dispatch_Id']); acc[k]=acc.get(k,0)+float(r['Counter_Value'])
for k in sorted(acc): print(k,acc[k])
PY
