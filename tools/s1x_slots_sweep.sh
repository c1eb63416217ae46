for s in 384 448 512 576 640; do
  for rep in 1 2; do
    v=$(CAELO_S1X_SLOTS=$s python bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    echo "slots $s rep $rep: $v"
  done
done
