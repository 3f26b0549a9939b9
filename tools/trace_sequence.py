import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last occurrence of sat_rows_pipe and print the sequence from there until next sat_rows
idx=[i for i,r in enumerate(rows) if 'sat_rows' in r['Kernel_Name']]
a=idx[-3]; b=idx[-2]
t0=int(rows[a]['Start_Timestamp'])
prev_end=t0
for r in rows[a:b]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:8.2f} gap {(s-prev_end)/1e3:6.2f} dur {(e-s)/1e3:6.2f}  {r['Kernel_Name'][:60]} grid {r.get('Grid_Size','')} wg {r.get('Workgroup_Size','')}")
    prev_end=e
print('total',(int(rows[b]['Start_Timestamp'])-t0)/1e3)
