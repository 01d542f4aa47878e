out=gpurun_out/r4n; mkdir -p $out
timeout -k 10 300 python -u -m pytest tests/test_gpu_fuse16w.py -x -q --timeout 120 --timeout-method thread > $out/t.log 2>&1 || { tail -5 $out/t.log; exit 1; }
tail -1 $out/t.log
for cfgname in "v128:--vertices 128" "v196:--vertices 196" "v256:--vertices 256" "cfg3:--workload cfg3"; do
  name=${cfgname%%:*}; flags=${cfgname#*:}
  for i in 1 2; do
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline $flags > $out/${name}_auto_$i.json 2>> $out/err.log || exit 1
    DANCE_FUSE16_SLOTS=0 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline $flags > $out/${name}_off_$i.json 2>> $out/err.log || exit 1
  done
done
