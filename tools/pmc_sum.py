import csv,sys,glob,collections
for d in sys.argv[1:]:
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:70]+' grid='+r.get('Grid_Size','')
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        for k,v in acc.items():
            print(k); 
            for c,x in sorted(v.items()): print(f"   {c:32s} {x:.4g}")
