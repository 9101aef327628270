import csv,sys,glob
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'scale_x' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
prev_end=0
for r in rows[idx:idx+60]:
    n=r['Kernel_Name'].split('(')[0].replace('void gpso::','')[:34]
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    gx=int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
    print(f"{s:9.1f} -> {e:9.1f}  q{r['Queue_Id']} s{r['Stream_Id']} {n:34s} grid({gx},{r['Grid_Size_Y']},{r['Grid_Size_Z']})")
