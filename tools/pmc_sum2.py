import csv,sys,glob,collections
# per-kernel averages of counters, filtered by substring
pat=sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                key=(r['Kernel_Name'][:30], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), r.get('Grid_Size_Y',''))
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print(k, {c: round(sum(x)/len(x),0) for c,x in v.items()}, 'n=',len(next(iter(v.values()))))
