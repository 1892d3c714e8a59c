# A/B of the fused kernels' knobs on one box (cold kernel table): rows per iteration, prefetch depth, twiddle storage, accumulation
for cfg in "" "FZ_FUSED_TW=1" "FZ_FUSED_TW=1 FZ_NO_IMAD=1" "FZ_FUSED_PREFETCH=2" "FZ_FUSED_ROWS=2" "FZ_FUSED_TW=1 FZ_FUSED_ROWS=2"; do
  echo "## ${cfg:-default (rows 1, prefetch 1, twiddle pairs, integer accumulation)}"
  env $cfg timeout -k 10 200 python tools/kernel_table.py --only fused 2>&1 | grep -E "keygen|G=8192|G=1024|polymul"
done
