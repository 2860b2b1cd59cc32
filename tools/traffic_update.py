"""Write profiles/traffic_latest.json from a tools/pmc_sq.sh output directory: HBM bytes of one ms_fwd_adj_kernel launch
(2 * FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) stamped with the identity of the kernel sources
and the batch size the passes were taken at.  bench.py only reports `roofline.traffic` from a file whose stamp matches.

    tools/pmc_sq.sh gpurun_out/pmc_B32 32 && python tools/traffic_update.py gpurun_out/pmc_B32 32 [label]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha, algorithmic_bytes_fwd_grad      # noqa: E402

out_dir, B = sys.argv[1], int(sys.argv[2])
label = sys.argv[3] if len(sys.argv) > 3 else out_dir
summ = json.load(open(os.path.join(out_dir, 'summary.json')))
k = [n for n in summ if 'ms_fwd_adj_kernel' in n and 'FETCH_SIZE' in summ[n] and 'WRITE_SIZE' in summ[n]]
if not k:
    raise SystemExit('no ms_fwd_adj_kernel entry with FETCH_SIZE and WRITE_SIZE in %s/summary.json' % out_dir)
# the measured launch is the LAST kernel instance listed for the forward+adjoint template (kbench launches only that one)
c = summ[k[-1]]
res = {
    'ms_fwd_adj_kernel_hbm_bytes_per_launch': 2 * c['FETCH_SIZE'] * 1024 + c['WRITE_SIZE'] * 1024,
    'FETCH_SIZE_KB': c['FETCH_SIZE'], 'WRITE_SIZE_KB': c['WRITE_SIZE'],
    'correction': 'gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM): read bytes = 2*FETCH_SIZE*1024; WRITE_SIZE is exact.',
    'command': 'tools/pmc_sq.sh (separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE) -- python3 tools/kbench.py %d 2  (config 3)' % B,
    'algorithmic_bytes_per_launch': algorithmic_bytes_fwd_grad(B, 72, 72, 256, 256 ** 3),
    'pmc_kernel_sources_sha': kernel_sources_sha(), 'pmc_batch': B, 'source': label,
}
json.dump(res, open(os.path.join(ROOT, 'profiles', 'traffic_latest.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
