"""Power / clock trace beside a running kernel loop: starts `python tools/kbench.py B REPS` as a child process and samples the
GPU's socket power and shader clock while it runs (hwmon sysfs files where readable -- ~1 ms per sample -- and
`rocm-smi --showpower --showclocks --json` every ~0.5 s as the cross-check), then prints a summary and writes the samples.

    python tools/power_trace.py [B=256] [reps=3000] [out.json]

Evidence for DESIGN.md section 5: at 256 positions the same instruction stream takes 2.55 ms against 1.80 ms at 32; cycles /
duration say 1.71 GHz against 2.37 GHz.  This records what the SMU reports while that happens."""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hwmon_files():
    out = {}
    for d in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
        for key, names in (('power_uw', ('power1_average', 'power1_input')), ('sclk_hz', ('freq1_input',)), ('mclk_hz', ('freq2_input',)),
                           ('temp_mc', ('temp1_input',)), ('power_cap_uw', ('power1_cap',))):
            for n in names:
                p = os.path.join(d, n)
                if key not in out and os.path.exists(p):
                    try:
                        int(open(p).read().strip())
                        out[key] = p
                    except Exception:
                        pass
        if out:
            break
    return out


def read_int(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None


def smi_sample():
    try:
        t = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10).stdout
        j = json.loads(t[t.index('{'):])
        card = j[sorted(j)[0]]
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ('power', 'sclk', 'mclk', 'fclk'))}
        return keep
    except Exception as e:
        return {'error': repr(e)}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'gpurun_out', 'power_trace_B%d.json' % B)
    files = hwmon_files()
    idle = {k: read_int(p) for k, p in files.items()}
    idle_smi = smi_sample()
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'kbench.py'), str(B), str(reps)], stdout=subprocess.PIPE, text=True)
    samples, smi = [], []
    t0 = time.time()
    next_smi = t0 + 2.0
    while child.poll() is None:
        now = time.time()
        s = {'t': now - t0}
        for k, p in files.items():
            if k != 'power_cap_uw':
                s[k] = read_int(p)
        samples.append(s)
        if now >= next_smi:
            q = smi_sample()
            q['t'] = time.time() - t0
            smi.append(q)
            next_smi = time.time() + 0.5
        time.sleep(0.02)
    kb = child.stdout.read().strip()
    summary = {'B': B, 'reps': reps, 'kbench': kb, 'hwmon_files': files, 'idle': idle, 'idle_smi': idle_smi, 'n_samples': len(samples)}
    # the loaded phase: the last 60 % of the run (the first part is import + setup)
    act = [s for s in samples if s['t'] > 0.4 * samples[-1]['t']] if samples else []
    for k in ('power_uw', 'sclk_hz', 'mclk_hz', 'temp_mc'):
        v = [s[k] for s in act if s.get(k) is not None]
        if v:
            v.sort()
            summary[k + '_loaded'] = {'median': v[len(v) // 2], 'p10': v[len(v) // 10], 'p90': v[(9 * len(v)) // 10], 'max': v[-1]}
    summary['smi_loaded'] = smi[len(smi) // 2:] if smi else []
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump({'summary': summary, 'samples': samples[::5], 'smi': smi}, open(out_path, 'w'), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == '__main__':
    main()
