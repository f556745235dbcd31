"""Distribution of the goal-set kernel's launch durations in a rocprofv3 --kernel-trace CSV: overall and per queue (= pipeline part), by the
bench's timing stride, and as a histogram in 10 us bins.  When the pipeline's halves drift apart (more often under the profiler), a half's
launch that runs alone takes ~115 us instead of ~170 us beside the other half's: the mean drops, the median does not.
    python tools/trace_goalset_durations.py <kernel_trace.csv>"""
import csv,sys,statistics as st
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_goalset_queue<2, false, false, false' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
q=sorted({r['Queue_Id'] for r in rows})
print('n',len(d),'mean',round(st.mean(d),1),'median',round(st.median(d),1))
for qq in q:
    dq=[x for x,r in zip(d,rows) if r['Queue_Id']==qq]
    print('queue',qq,len(dq),'mean',round(st.mean(dq),1),'median',round(st.median(dq),1),'p10',round(sorted(dq)[len(dq)//10],1),'p90',round(sorted(dq)[9*len(dq)//10],1))
# every 5th dispatch in launch order per the bench's stride
for off in range(5):
    s=d[off::5]; print('stride5 offset',off,round(st.mean(s),1))
# histogram
import collections
h=collections.Counter(int(x//10)*10 for x in d)
print(sorted(h.items()))
