import csv,sys,glob,collections
# sum counter values per kernel name prefix over dispatches
for d in sys.argv[1:]:
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:40]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
            n[(k,r['Counter_Name'])]+=1
        for k,v in acc.items():
            if 'gl_iter' in k:
                print(k, {c: round(x/n[(k,c)],1) for c,x in v.items()})
