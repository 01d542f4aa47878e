timeout -k 10 300 python -u -m pytest tests/test_gpu_snake.py tests/test_gpu_modules.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
for rep in 1 2; do for st in 0 1; do
DANCE_RES_STACK=$st timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-feat-prep > /tmp/b.json 2>/dev/null; echo "stack=$st"; python tools/show_bench.py /tmp/b.json
done; done
