"""The driver in update_scheme='per angle' on config 3's shape (6 first-touch angles), as a program a profiler can wrap:
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/driver_per_angle_probe.py ; python tools/trace_tail.py OUT/*/*kernel_trace.csv 64"""
import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from adorym_amd import workloads as W
print(bench.driver_measure(W.c3_config(), 6, 'per angle'))
