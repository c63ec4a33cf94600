for rep in 1 2 3; do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline | python scripts/ab_fmt.py "$cfg"
  done
done
