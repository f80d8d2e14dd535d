#!/bin/bash
# instruction mix of the funnel workload's kernel (run on the GPU box): tools/pmc_funnel.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_funnel; rm -rf $O; mkdir -p $O; cd $R
A="--workload funnel --no-extras --no-cpu-baseline --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $O/p1 -- python3 bench.py $A > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU_IOPS SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $O/p2 -- python3 bench.py $A > $O/p2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $O/p3 -- python3 bench.py $A > $O/p3.log 2>&1
python3 - <<'PY'
import csv,glob,os
acc={}
for f in glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_funnel/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wn_wave' in r['Kernel_Name'] or 'wn_orbit' in r['Kernel_Name']:
            k=r['Counter_Name']; acc[k]=acc.get(k,0)+float(r['Counter_Value'])
E=1.204e8  # gradient evaluations of one launch of the bench workload (4096 chains x 300 iterations)
for k in sorted(acc): print(f"{k:28s} {acc[k]:.4g}  per evaluation {acc[k]/E:.1f}")
PY
