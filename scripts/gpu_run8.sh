cd /root/repo
for seed in 1 2 3 5 7 11; do python scripts/fuzz_reuse.py 250 $seed 2>&1 | tail -1; done
for seed in 1 2 3; do python scripts/fuzz_parity.py 150 $seed 2>&1 | tail -1; done
