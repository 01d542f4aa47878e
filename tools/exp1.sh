KB_ALL=1 timeout -k 10 60 python tools/kbench16.py 256 512 1600 2>&1 | grep -v amdgpu.ids
