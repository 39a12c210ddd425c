cd /root/repo
for it in 5 20 60 200 520; do BF_DEBUG_MARGIN=1 python scripts/kernel_time.py 260 346 bin_split=2 co_schedule=1 iters=$it 2>&1 | grep -E "margin|K1" | tail -2; done
BF_DEBUG_MARGIN=1 python scripts/kernel_time.py 480 640 bin_split=2 co_schedule=0 iters=100 2>&1 | grep -E "margin|K1" | tail -2
