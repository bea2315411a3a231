"""Print per-kernel PMC counters (last dispatch of each kernel name) from rocprofv3 csv passes."""
import csv, collections, sys
base = sys.argv[1]
for name in sys.argv[2:]:
    try:
        rows = list(csv.DictReader(open(f'{base}/{name}/p_counter_collection.csv')))
    except FileNotFoundError:
        print('==', name, 'missing'); continue
    d = collections.OrderedDict()
    for r in rows:
        if 'pack' in r['Kernel_Name']: continue
        k = r['Kernel_Name'][:60]
        d.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])   # keeps the last dispatch
    for k, v in d.items():
        print('==', name, k, {a: f'{b:.4g}' for a, b in v.items()})
