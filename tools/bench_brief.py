"""One-line summaries of bench.py JSON lines (files given on the command line): headline value, step and kernel time, and the
full-chip legs.  python tools/bench_brief.py gpurun_out/a.json gpurun_out/b.json"""
import json
import sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e:
        print(f, 'unreadable:', repr(e))
        continue
    s = '%s: %.0f pos/s, step %.3f ms, kernel %.3f ms, loss %.4f' % (f, j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['loss_last'])
    for k in ('per_angle', 'virtual_ranks_8', 'virtual_ranks_16'):
        if k in j:
            s += ' | %s %.3f ms (launch part %.3f, %.1f %%)' % (k, j[k]['ms_per_step'], j[k]['fwd_adj_overlap_add_ms'], 100 * j[k]['whole_step_frac_of_hbm_peak'])
    print(s)
