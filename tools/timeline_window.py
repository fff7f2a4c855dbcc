import csv,glob,sys
rows=list(csv.DictReader(open(glob.glob('/tmp/qt/**/*kernel_trace.csv',recursive=True)[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
key=sys.argv[1]
ref=[i for i,r in enumerate(rows) if key in r['Kernel_Name']]
i0=ref[len(ref)//3]; t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    if s>float(sys.argv[2]): break
    n=r['Kernel_Name'].replace('void ','').replace('ftrl_dev::','')
    if 'rocprim' in n: n='sort:'+n.split('detail::')[-1][:20]
    print('%8.1f %8.1f %7.1f q=%s %s'%(s,e,e-s,r['Queue_Id'],n[:34]))
