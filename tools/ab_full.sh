echo "== default"; python tools/time_full.py 2>&1 | tail -2
echo "== THA4_TILE_1X1"; THA4_TILE_1X1=1 python tools/time_full.py 2>&1 | tail -2
