#!/bin/bash
# A/B of the many-aggregates verification kernels on cold operands (tools/kernel_table.py --only verify), one box, one run.
# FZ_VERIFY16: 0 = the radix-4 kernel (verify_fused), 1 .. 6 = the 16-per-lane kernel (verify_many16) with that many
# waves per workgroup, 7 = with the divisor of the wave-tasks.  FZ_VERIFY16_NOPF=1: rows requested when the task starts instead of
# one task ahead (fewer registers, more waves).  Applies from 512 aggregates per launch on.
set -e
cd "$(dirname "$0")/.."
for sp in 256 128; do
  for s in 0 1 2 3 4 6 7; do
    for n in 0 1; do
      [ "$s" = 0 ] && [ "$n" = 1 ] && continue
      echo "== secpar $sp FZ_VERIFY16=$s FZ_VERIFY16_NOPF=$n"
      FZ_VERIFY16=$s FZ_VERIFY16_NOPF=$n python tools/kernel_table.py --only "verify_fused G=" --secpar $sp 2>&1 | grep -E "G=1024|G=8192"
    done
  done
done
