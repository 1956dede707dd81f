cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s37; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
