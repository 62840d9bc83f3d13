for k in 16 8 4 2; do
  echo "== THA4_KSPLIT_MAX=$k"; THA4_KSPLIT_MAX=$k python tools/time_full.py 2>&1 | tail -2
done
