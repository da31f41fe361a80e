"""Composition of one hipGraph-replayed rollout step from a rocprofv3 --kernel-trace of tools/bench_rollout.py (graph mode):
mean duration per kernel class and per step over the last rollout.  usage: tools/graph_step_gaps.py <trace directory>"""
import csv,glob,statistics,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'policy_act' in n]
per=[]
for a,b in zip(idx[-64:-1], idx[-63:]):
    t0=int(rows[a]['Start_Timestamp']); t1=int(rows[b]['Start_Timestamp'])
    d={}
    for r in rows[a:b]:
        k=[x for x in ('policy_act','preamble','solve_wave','synth_env','rollout_record') if x in r['Kernel_Name']]
        k=k[0] if k else 'torch'
        d[k]=d.get(k,0)+(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000
    d['period']=(t1-t0)/1000; d['n']=b-a
    per.append(d)
keys=sorted({k for d in per for k in d})
for k in keys: print(k, round(statistics.mean(d.get(k,0) for d in per),1))
print('non-solve', round(statistics.mean(d['period']-d['solve_wave'] for d in per),1))
