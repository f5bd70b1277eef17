import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    n=r['Name']
    if 'node::' in n:
        print('   %-44s calls %5s avg_us %8.2f'%(n.split('(')[0][:44], r['Calls'], float(r['AverageNs'])/1e3))
