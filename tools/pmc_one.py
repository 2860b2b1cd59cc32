"""Print the counters of the LAST dispatch of every kernel whose name contains <substr> from the CSVs under <dir>
(written by `rocprofv3 --pmc ... --output-format csv -d <dir> -- python3 ...`).  python tools/pmc_one.py <dir> <substr>"""
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name']]
    if not rows:
        continue
    last = max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id']) == last:
            print(r['Counter_Name'], r['Counter_Value'], 'vgpr', r.get('VGPR_Count'), 'grid', r.get('Grid_Size'))
