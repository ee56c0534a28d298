cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for Q in 384 1024; do
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_rq_$Q -o p -- python3 bench.py --batch-queries $Q --k 30 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_rq_$Q.err || { tail -5 gpurun_out/pmc_rq_$Q.err; exit 1; }
python3 - $Q <<'PY'
import csv,glob,sys
Q=sys.argv[1]
f=glob.glob(f'gpurun_out/pmc_rq_{Q}/**/*counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'rq16' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
import collections
per=collections.defaultdict(float)
for r in rows: per[r['Dispatch_Id']]+=float(r['Counter_Value'])
vals=sorted(per.values())
print('Q',Q,'rq16 launches',len(vals),'FETCH_SIZE x1024 x2 per launch GB:',[round(v*1024*2/1e9,2) for v in vals][:6])
PY
done
