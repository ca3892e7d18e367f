python -m pytest tests/test_gpu_round6.py tests/test_gpu_lattn_fused.py tests/test_gpu_ops.py -q -m gpu -x -k "stem or lattn or merge or linattn" 2>&1 | tail -4
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --max-seconds 100000 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step', b['ms_per_step'])
print(json.dumps(b.get('sampling'), indent=0)[:1500])"
