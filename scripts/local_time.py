import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from better_flow_amd import accel, synth
N, H, W, s = 1000000, 260, 346, 3
sl = synth.make_slice(N, H, W, 0.030, seed=1)
acc = accel.Accel(max_events=N, max_rows=s*H+s, max_cols=s*W+s)
acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
acc.local_set_window(s)
acc.local_iteration_step(0, 0)
t0 = time.perf_counter(); n = 50
for i in range(n): acc.local_iteration_step(0.01 * i, 0.0)
dt = (time.perf_counter() - t0) / n
print("local score evaluation: %.1f us (%.1f Gev/s)" % (dt * 1e6, N / dt / 1e9))
t0 = time.perf_counter(); rc, st = acc.local_run(H, W); dt = time.perf_counter() - t0
print("local_run rc", rc, "evals", st.evaluations, "nx ny", st.nx, st.ny, "score", st.last_score, "ms %.2f" % (dt * 1e3))
