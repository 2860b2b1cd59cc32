"""Process-wide switches, mirroring adorym/global_settings.py:1-5.  The only backend is 'hip'."""
backend = 'hip'
xpu = False
run_bf16 = False
run_fp64 = False
