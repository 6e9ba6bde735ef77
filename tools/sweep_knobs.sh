for cfg in "" "CCR_PROGRESSIVE=0" "CCR_SAMPLE_DIV=32" "CCR_SAMPLE_DIV=32 CCR_PROGRESSIVE=0" "CCR_SAMPLE_DIV=16 CCR_PROGRESSIVE=0" "CCR_QGROUPS=1" "CCR_QGROUPS=1 CCR_PROGRESSIVE=0" "CCR_MFMA16=0"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 5 --no-secondary --cpu-queries 0 2>&1 | grep -E "timed|warmup step" | tail -2 | sed 's/.*ranges/ranges/' | cut -c1-260
done
