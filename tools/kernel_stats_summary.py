import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=[int(r['Calls']) for r in rows if 'adam' in r['Name']][0]
print("steps",steps,"launches/step",sum(int(r['Calls']) for r in rows)/steps,"us/step",tot/steps/1e3)
for r in rows[:12]:
    print(f"  {r['Name'][:80]:80s} calls/step {int(r['Calls'])/steps:6.1f} avg_us {float(r['AverageNs'])/1e3:7.2f} pct {float(r['TotalDurationNs'])/tot*100:5.1f}")
