import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from adorym_amd import workloads as W
print(bench.driver_measure(W.c3_config(), 6, 'per angle'))
