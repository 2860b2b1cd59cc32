"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py gpurun_out/prof_x"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print('%-44s calls %3s avg %9.1f us  min %9.1f  max %9.1f' % (r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                   float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
