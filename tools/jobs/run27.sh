cd $GRAFT_REPO_ROOT
T=tests/test_gpu_parity.py::test_rift_loss_of_the_full_fixture_in_fp16
for cfg in "1 1" "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
echo "== NAT_MAIN=$1 COMPACT=$2"; RIFT_NAT_MAIN=$1 RIFT_NAT_COMPACT=$2 timeout 600 python -m pytest $T -x -q -m gpu -s 2>&1 | grep "reference\|passed\|failed"
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropstats.py tests/test_gpu_shapes.py -q -m gpu 2>&1 | tail -8
