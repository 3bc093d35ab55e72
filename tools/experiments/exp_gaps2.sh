cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -o g -- python $R/bench.py --no-cpu-baseline --steps 60 --warmup 10 $BENCH_ARGS > /dev/null 2>&1
python - <<'P'
import csv,glob,os,statistics
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/gaps/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
at=[r for r in rows if 'atrous' in r['Kernel_Name']]
at.sort(key=lambda r:int(r['Start_Timestamp']))
tp=[r for r in rows if 'temporal' in r['Kernel_Name']]
ks=[(r['Kernel_Name'][29:52],int(r['Start_Timestamp']),int(r['End_Timestamp']),r.get('Queue_Id','?'),r.get('Stream_Id','?')) for r in at]
# steady-state part: last 150 atrous kernels
sel=ks[-150:]
gaps=[(sel[i][1]-sel[i-1][2])/1000 for i in range(1,len(sel))]
durs=[(k[2]-k[1])/1000 for k in sel]
for i in range(len(sel)-15,len(sel)):
    print(sel[i][0], 'dur %.1f'%durs[i], 'gap %.2f'%(gaps[i-1]), 'queue',sel[i][3],sel[i][4])
print('median gap %.2f us, mean gap %.2f, median dur %.1f, mean dur %.1f'%(statistics.median(gaps),statistics.mean(gaps),statistics.median(durs),statistics.mean(durs)))
span=(sel[-1][2]-sel[0][1])/1000
print('span of 150 levels %.1f us -> %.1f us per frame of 5 levels; sum dur %.1f, sum gaps %.1f'%(span,span/30,sum(durs),sum(gaps)))
td=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in tp][-30:]
print('temporal median dur %.1f'%statistics.median(td))
P
rm -rf $R/gpurun_out/gaps
