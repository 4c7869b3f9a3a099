cd $GRAFT_REPO_ROOT
T="tests/test_gpu_properties.py::test_eval_forward_is_scene_permutation_equivariant_at_benchmark_size"
for e in "X=1" "RIFT_NAT_COMPACT=0" "RIFT_PE_W=0"; do echo "== $e"; env $e timeout 600 python -m pytest "$T" -q -m gpu 2>&1 | tail -2; done
