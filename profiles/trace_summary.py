import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'][:40]; g=int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
    agg[(n,g,r['LDS_Block_Size'],r['VGPR_Count'],r['Scratch_Size'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    v=sorted(v); print(k, 'n=%d med=%.1fus min=%.1f max=%.1f'%(len(v), v[len(v)//2], v[0], v[-1]))
