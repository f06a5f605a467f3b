for rep in 1 2 3; do
  for lib in libnmhip_old.so libnmhip.so; do
    echo -n "$lib: "
    NMHIP_LIB_NAME=$lib python bench.py --cpu-budget 0 --small-sweep 0 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
