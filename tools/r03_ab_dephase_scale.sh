for rep in 1 2; do
  for sc in 0 0.5 1 2 4; do
    echo -n "scale $sc: "
    NMHIP_DEPHASE_SCALE=$sc python bench.py --cpu-budget 0 --small-sweep 0 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
