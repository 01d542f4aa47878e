out=gpurun_out/r4l; mkdir -p $out
for cfgname in "cfg3:--workload cfg3" "v196:--vertices 196" "v256:--vertices 256"; do
  name=${cfgname%%:*}; flags=${cfgname#*:}
  for i in 1 2; do
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline $flags > $out/${name}_peel_$i.json 2>> $out/err.log || exit 1
    DANCE_PEEL=0 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline $flags > $out/${name}_one_$i.json 2>> $out/err.log || exit 1
  done
done
