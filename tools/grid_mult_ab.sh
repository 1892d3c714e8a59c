for g in 1 2 4 8; do echo "## FZ_NTT_GRID_MULT=$g"; FZ_NTT_GRID_MULT=$g timeout -k 10 200 python tools/kernel_table.py --only "B=2^1" 2>&1 | grep -E "2\^16|2\^18"; done
