run() { name=$1; shift; "$@" > gpurun_out/r05a/bis3_$name.log 2>&1; echo "$name rc=$?"; tail -2 gpurun_out/r05a/bis3_$name.log; }
run full1 python -X faulthandler -m pytest tests -m gpu -x -q
run full_torchfirst env ADM_TEST_TORCH_FIRST=1 python -X faulthandler -m pytest tests -m gpu -x -q
run no_world2 python -X faulthandler -m pytest tests -m gpu -x -q --deselect tests/test_gpu_world2.py
