import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][28:80], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
