cd /root/repo
for m in 0 1; do
echo "co_schedule (tail) = $m"; BF_CO=$m python scripts/timeline_k3.py 2>&1 | tail -8
done
