import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void ','')) for r in csv.DictReader(open(f))]
rows.sort()
rows=rows[len(rows)//2:]   # steady part
dur=collections.defaultdict(list); gap=collections.defaultdict(list)
for (s0,e0,n0),(s1,e1,n1) in zip(rows,rows[1:]):
    dur[n0].append(e0-s0); gap[(n0[:28],n1[:28])].append(s1-e0)
print('kernel durations (us): n avg')
for n,v in sorted(dur.items(), key=lambda kv:-sum(kv[1]))[:14]: print('  %-40s %6d %8.2f'%(n[:40],len(v),sum(v)/len(v)/1e3))
print('gaps (us): n avg')
for k,v in sorted(gap.items(), key=lambda kv:-sum(kv[1]))[:12]: print('  %-28s -> %-28s %6d %8.2f'%(k[0],k[1],len(v),sum(v)/len(v)/1e3))
tot=rows[-1][1]-rows[0][0]; busy=sum(e-s for s,e,_ in rows)
print('span %.1f us, busy %.1f us (%.0f%%), launches %d, span/launch %.2f us'%(tot/1e3,busy/1e3,100*busy/tot,len(rows),tot/1e3/len(rows)))
