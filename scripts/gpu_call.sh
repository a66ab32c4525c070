cd /root/repo
O=gpurun_out/r03_y; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
tail -3 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
