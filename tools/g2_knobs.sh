for cfg in "16 8 6" "16 8 0" "8 8 6" "24 8 6" "32 8 6" "16 16 6" "16 4 6" "16 8 12" "16 0 0"; do
  set -- $cfg
  echo "refill_min=$1 tail_lanes=$2 tail_age=$3: random $(ABD_G2_REFILL_MIN=$1 ABD_G2_TAIL_LANES=$2 ABD_G2_TAIL_AGE=$3 python tools/probe_gibbs.py 10 | tail -1)  truth $(ABD_G2_REFILL_MIN=$1 ABD_G2_TAIL_LANES=$2 ABD_G2_TAIL_AGE=$3 python tools/probe_gibbs.py 10 truth | tail -1)"
done
