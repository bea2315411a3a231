import csv, collections, sys
def load(path):
    rows = list(csv.DictReader(open(path)))
    d = collections.OrderedDict()
    for r in rows:
        k = (r['Dispatch_Id'], r['Kernel_Name'])
        d.setdefault(k, {'_t': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})[r['Counter_Name']] = float(r['Counter_Value'])
    return d
base = sys.argv[1]
for name in sys.argv[2:]:
    d = load(f'{base}/{name}/p_counter_collection.csv')
    items = [(k, v) for k, v in d.items() if 'pack' not in k[1]]
    print('==', name, len(items))
    for ci in range(len(items) // 6):
        k, v = items[ci * 6 + 3]
        print(ci, k[1][11:45], {a: (f'{b:.3g}') for a, b in v.items()})
