for kn in "0.7,0.45,0.5,0.0,0.04" "0.7,0.45,0.5,0.07,0.0" "0.7,0.45,0.0,0.0,0.0" "0.0,0.0,0.5,0.07,0.04" "0.0,0.0,0.0,0.0,0.0"; do
  for mode in exact plain calibrated; do
    echo "knobs $kn mode $mode: $(python scripts/tap_debug.py --weights trained --mode $mode --batch 8 --knobs $kn --brief 2>&1 | grep features)"
  done
done
