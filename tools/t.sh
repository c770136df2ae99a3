python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/quick_bench.py 512,1024,2048,4096
for r in 1 2 3; do python tools/depth_batch.py 2048 1 1,2,3; done
python tools/depth_batch.py 512 16 1,2
