DANCE_DEBUG_STAMPS=1 timeout -k 10 60 python tools/kbench16.py 2 1600 2>&1 | grep -v amdgpu.ids
