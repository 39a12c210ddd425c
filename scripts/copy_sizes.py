import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel
acc = accel.Accel(max_events=4096, max_rows=64, max_cols=64)
for mb in (4, 8, 16, 32, 64, 256, 1024):
    print(mb, "MB copy:", acc.copy_bandwidth(mb << 20, 4), "GB/s (event-timed, read+write)")
