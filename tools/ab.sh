# same-box A/B of two builds of libdance_hip.so: bash tools/ab.sh <libA> <libB>
for rep in 1 2; do
  for L in $1 $2; do
    DANCE_HIP_LIB=$(pwd)/$L timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /tmp/ab.json 2>/dev/null && echo "$L" && python tools/show_bench.py /tmp/ab.json
  done
done
