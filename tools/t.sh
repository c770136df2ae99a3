python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/quick_bench.py 2048,4096
for r in 1 2; do python tools/depth_batch.py 2048 1 1,2,3,4; done
python tools/depth_batch.py 4096 1 1,2
