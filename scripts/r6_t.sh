cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python scripts/h2h_warm.py 1 ahead=2 defer=1
python scripts/fuzz_reuse.py 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()"
