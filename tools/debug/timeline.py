import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_rollout' in r['Kernel_Name']]
i0=idx[80]; t0=int(rows[i0]['Start_Timestamp'])
import re
for r in rows[i0-3:idx[82]+1]:
    m=re.search(r'(k2?_[a-z_0-9]+)', r['Kernel_Name'])
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-t0)/1e3:8.1f}  q{r['Queue_Id']}  {m.group(1) if m else r['Kernel_Name'][:20]}")
