import csv,glob,sys
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if 'cem_' in r['Name']:
            print(d, r['Name'][:60].replace('(anonymous namespace)::',''), 'calls', r['Calls'], 'avg us %.1f'%(float(r['AverageNs'])/1e3))
