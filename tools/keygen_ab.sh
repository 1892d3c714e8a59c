for cfg in "" "FZ_FUSED_PREFETCH=1" "FZ_FUSED_ROWS=2" "FZ_FUSED_ROWS=2 FZ_FUSED_PREFETCH=1"; do
  echo "## ${cfg:-default (rows 1, prefetch 2)}"
  env $cfg timeout -k 10 200 python tools/kernel_table.py --only keygen 2>&1 | grep keygen
done
