"""Summarise rocprofv3 --pmc passes written by tools/pmc_sq.sh: per counter, the value of the LAST dispatch of every
kernel whose name contains ms_ (forward+adjoint launch of kbench)."""
import sys, glob, csv, collections, json
out = sys.argv[1]
res = collections.OrderedDict()
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    rows = list(csv.DictReader(open(f)))
    byk = collections.defaultdict(dict)
    for r in rows:
        k = r.get('Kernel_Name', '')
        if 'ms_' not in k:
            continue
        k = k.split('(')[0][:60]
        byk[k].setdefault(r['Counter_Name'], []).append((int(r['Dispatch_Id']), float(r['Counter_Value']), r.get('VGPR_Count'), r.get('LDS_Block_Size')))
    for k, d in byk.items():
        for c, v in d.items():
            v.sort()
            res.setdefault(k, {})[c] = v[-1][1]
            res[k]['_vgpr'] = v[-1][2]
            res[k]['_lds'] = v[-1][3]
print(json.dumps(res, indent=1))
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
