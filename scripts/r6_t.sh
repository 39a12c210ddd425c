cd $GRAFT_REPO_ROOT
A=$PWD/better_flow_amd/libbf_accel_alt.so; B=$PWD/better_flow_amd/libbf_accel.so
bash scripts/ab_k3.sh $A $B 2
for i in 1 2; do for L in $A $B; do echo "$(basename $L) $(BF_ACCEL_LIB=$L python bench.py --steps 20 --no-cpu-baseline --no-front-end 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['roofline']['chip_full']['stencil_us_per_config2_image'])")"; done; done
for L in 3 4 5 6; do echo "lanes $L: $(python bench.py --config 5 --farm-slices 24 --concurrent $L 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])")"; done
