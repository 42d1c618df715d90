import csv, sys, re, glob
from collections import defaultdict
csv.field_size_limit(1<<30)
agg=defaultdict(lambda: defaultdict(list))
for p in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").replace("zkp::","")[:60]
        if "at::" in k or "rocclr" in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(agg.items()):
    print(k, {c: round(sum(v)/len(v),1) for c,v in sorted(cs.items())})
