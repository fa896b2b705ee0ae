timeout 300 python -m pytest tests/test_gpu_stage2.py -x -q -m gpu 2>&1 | tail -15
run() { echo "== $*"; timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; }
run --contexts 1
run --contexts 2
run --contexts 3
run --contexts 4
run --contexts 3 --walk steps
