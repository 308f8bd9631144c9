import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
sel=[r for r in rows if 'sw2d' in r['Kernel_Name'] or 'nccl' in r['Kernel_Name'] or 'halo' in r['Kernel_Name']]
n=len(sel)
for r in sel[n-40:n-16]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{s:10.1f} {d:7.1f} {s+d:10.1f}  {r['Kernel_Name'][:70]}")
