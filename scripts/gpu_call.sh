cd /root/repo
O=gpurun_out/r03_z2; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py::test_prune_on_the_downdate_equals_the_separate_prune -m gpu -x -q 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest.txt
