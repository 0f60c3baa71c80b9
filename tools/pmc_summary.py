"""Summarise rocprofv3 --pmc CSVs written by tools/pmc.sh: one line per search-kernel dispatch."""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

out = sys.argv[1]
table = OrderedDict()
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_search" not in k and "k_contested" not in k and "k_finalize" not in k:
            continue
        key = (int(r["Dispatch_Id"]), k.split("(")[0][-40:])
        table.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
# dispatch ids differ between passes only if the program differs; group by order of appearance
names = []
for key in table:
    for c in table[key]:
        if c not in names:
            names.append(c)
print("dispatch kernel " + " ".join(names))
for key, vals in table.items():
    print(key[0], key[1], " ".join("%s=%.4g" % (n, vals.get(n, float("nan"))) for n in names))
