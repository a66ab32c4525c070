cd /root/repo
O=gpurun_out/r03_z3; mkdir -p $O
timeout 110 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest.txt
