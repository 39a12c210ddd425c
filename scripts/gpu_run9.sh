cd /root/repo
python - <<'PY'
import sys, time; sys.path.insert(0,'.')
import numpy as np
from better_flow_amd import synth
n = synth.write_stream_bin('/tmp/t.bin', 4, 1000000, 260, 346)
raw = np.fromfile('/tmp/t.bin', dtype=np.uint8)
t = raw[16:16+8*n].view('<u8'); x = raw[16+8*n:16+10*n].view('<u2'); y = raw[16+10*n:16+12*n].view('<u2')
t0=time.time()
with open('/tmp/t.txt','w') as f:
    f.write(''.join('%.9f %d %d 1\n' % (a*1e-9, b, c) for a, b, c in zip(t.tolist(), x.tolist(), y.tolist())))
print('text file', n, 'events', time.time()-t0, 's')
PY
ls -la /tmp/t.txt
F="--quiet --timing --res-x=260 --res-y=346 --max-events=1100000 --span=0.03 --refresh-time=0.03 --refresh-event-count=1000000000"
for th in 1 4 8 16; do better_flow_amd/host/bf_motion_compensator $F --threads=$th /tmp/t.txt 2>&1 | tail -1 | cut -c1-330; done
better_flow_amd/host/bf_motion_compensator $F --threads=8 -o /tmp/o.txt /tmp/t.txt 2>&1 | tail -1 | cut -c1-330
better_flow_amd/host/bf_motion_compensator $F --engine=ring -o /tmp/o2.txt /tmp/t.bin 2>&1 | tail -2 | cut -c1-200
