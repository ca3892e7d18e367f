timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv" 2>&1 | tail -3
timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tail -14
