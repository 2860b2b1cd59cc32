"""Which kernels run while the longest memory copies of a rocprofv3 --kernel-trace --memory-copy-trace run are in flight.
python tools/copy_overlap.py <dir> [n]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cp = list(csv.DictReader(open(glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)[0])))
ks = sorted(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])), key=lambda r: int(r['Start_Timestamp']))
cp.sort(key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for r in sorted(cp[-n:], key=lambda r: int(r['Start_Timestamp'])):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ov = [x['Kernel_Name'].split('(')[0][-36:] for x in ks if int(x['Start_Timestamp']) < e and int(x['End_Timestamp']) > s]
    print('%s %8.1f us at %.3f ms, stream %s, beside: %s' % (r['Direction'], (e - s) / 1e3, (s - int(ks[0]['Start_Timestamp'])) / 1e6, r['Stream_Id'], ov[:3] or 'nothing'))
