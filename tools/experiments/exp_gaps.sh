cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -o g -- python $R/tools/probe.py --variants 0 --frames 8 > /dev/null 2>&1
python - <<'P'
import csv,glob,os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/gaps/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[(r['Kernel_Name'][:48],int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows]
# last 40 kernels: print duration and gap to previous
import statistics
gaps=[];durs=[]
for i in range(len(ks)-60,len(ks)):
    n,s,e=ks[i]; ps=ks[i-1]
    gap=s-ps[2]
    print(f"{n:48s} dur {((e-s)/1000):7.1f} us  gap_to_prev_end {gap/1000:7.2f} us")
P
rm -rf $R/gpurun_out/gaps
