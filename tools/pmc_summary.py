"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel dispatch."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in acc for c in acc[k]})
print(f"{'kernel':48s} " + " ".join(f"{n[-14:]:>14s}" for n in names))
for k, d in sorted(acc.items(), key=lambda kv: -max(sum(v) for v in kv[1].values())):
    if 'at::' in k or 'rocclr' in k or 'rocprim' in k: continue
    print(f"{k:48s} " + " ".join(f"{(sum(d[n])/len(d[n]) if n in d else 0):14.0f}" for n in names))
