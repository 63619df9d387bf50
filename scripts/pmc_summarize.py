import csv, sys, glob, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        for k, c in acc.items():
            if "rt::" not in k: continue
            print(d, k, {a: (int(b) if b == int(b) else b) for a, b in c.items()})
