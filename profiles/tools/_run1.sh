timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), round(d['roofline']['frac'],4)); print({k:(round(v['value']/1e6,1), round(v['ms_per_step'],4), round(v['roofline']['frac'],4)) for k,v in d['also'].items()})"
