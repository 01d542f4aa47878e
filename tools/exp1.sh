for rep in 1 2; do
timeout -k 10 60 python tools/kbench16.py 1600 3200 2>&1 | grep -v amdgpu.ids
DANCE_MFMA_SHAPE=16 timeout -k 10 60 python tools/kbench16.py 1600 3200 2>&1 | grep -v amdgpu.ids
done
DANCE_MFMA_SHAPE=16 timeout -k 10 300 python -u -m pytest tests/test_gpu_snake.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
