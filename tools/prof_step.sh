cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof
B="python bench.py --workload cfg2 --steps 224 --warmup 112 --launch per_step --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/kt -o kt -- $B > gpurun_out/prof/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/prof/pmc1 -o pmc1 -- $B > gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES -d gpurun_out/prof/pmc2 -o pmc2 -- $B > gpurun_out/prof/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc3 -o pmc3 -- $B > gpurun_out/prof/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pmc4 -o pmc4 -- $B > gpurun_out/prof/pmc4.log 2>&1
find gpurun_out/prof -name "*.csv" | head -30
python - <<'P'
import csv,glob,collections
for f in glob.glob('gpurun_out/prof/kt/**/*kernel_stats.csv',recursive=True):
    print(open(f).read()[:3000])
for d in ['pmc1','pmc2','pmc3','pmc4']:
    for f in glob.glob(f'gpurun_out/prof/{d}/**/*counter_collection.csv',recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            print(d,k,{c:(sum(x)/len(x),len(x)) for c,x in v.items()})
P
