import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find multislice fwd_adj kernels; print timeline between the 20th and 24th
idx=[i for i,r in enumerate(rows) if 'ms_fwd_adj_kernel' in r['Kernel_Name']]
a=idx[int(sys.argv[2])]; b=idx[int(sys.argv[2])+3]
rows=rows[a-6:b+1]
t0=int(rows[0]['Start_Timestamp']); prev=t0
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=r['Kernel_Name']; n=n[:n.index('(')] if '(' in n else n
    print('%9.1f us  dur %8.1f  gap %7.1f  q%-3s grid %-8s %s'%((s-t0)/1e3,(e-s)/1e3,(s-prev)/1e3,r.get('Queue_Id','?'),r.get('Grid_Size_X',r.get('Grid_Size','?')),n[-60:]))
    prev=max(prev,e)
