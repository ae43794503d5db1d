for v in 0 1; do
  echo "== RAGMI_X3P=$v"
  RAGMI_X3P=$v python bench.py --no-configs --no-cpu-baseline 2>&1 | grep -E "conv3d x3 .*groups=\((2|3),\)|^\{" | sed 's/"config".*//' | cut -c1-230
  RAGMI_X3P=$v python bench.py --no-configs --no-cpu-baseline --batch 8 --steps 5 2>&1 | grep -E "conv3d x3 .*groups=\((2|3),\)|^\{" | sed 's/"config".*//' | cut -c1-230
done
