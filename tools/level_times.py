import csv,re,collections,sys
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    m=re.search(r'(block_kernel<[^>]*>|plane_kernel<[^>]*>|sine_solve_kernel<[^>]*>|tail_kernel<[^>]*>)',n)
    if m:
        acc[(m.group(1),r['Grid_Size_X'], r['Workgroup_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(acc.items()):
    print("  %-54s grid %-8s wg %-4s n %3d avg %7.1f us" % (k[0],k[1],k[2],len(v),sum(v)/len(v)))
