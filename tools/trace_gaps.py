import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40 kernels
tail=rows[-36:]
t0=int(tail[0]['Start_Timestamp'])
prev_end=None
for r in tail:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    name=r['Kernel_Name'].split('(')[0].split('::')[-1][:50]
    print(f"{(s-t0)/1e3:9.2f} dur {(e-s)/1e3:8.2f} gap {((s-prev_end)/1e3 if prev_end else 0):7.2f} {name} grid={r.get('Grid_Size_X','')} wg={r.get('Workgroup_Size_X','')}")
    prev_end=e
