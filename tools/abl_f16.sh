for v in 2 0 8 9; do
  for pr in 1 10 12; do
    echo -n "variant $v prio $pr: "
    LHG_GGS_VARIANT=$v LHG_GG_PRIO=$pr python tools/time_layer.py 512 256 96 3 1 fp32_split_f16 20 2>&1 | tail -1
  done
done
