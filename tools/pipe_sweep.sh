for cfg in "3 1" "2 2" "2 1" "4 1" "3 2" "6 1"; do
  set -- $cfg
  r=$(ABD_PIPES=$1 ABD_PIPE_BLOCKS_PER_CU=$2 python bench.py --no-cpu-baseline --steps 400 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_us'])")
  echo "pipes=$1 wg/cu=$2: $r"
done
