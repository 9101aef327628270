import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'leaf_tiles_v2' in r['Kernel_Name']]
i=idx[-3]
t0=int(rows[i-1]['Start_Timestamp'])
for r in rows[i-2:i+8]:
    n=r['Kernel_Name'].split('(')[0].replace('void gpso::','')[:40]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:9.1f}  {n}")
