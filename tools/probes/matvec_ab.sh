#!/bin/bash
# A/B of the matvec kernels on cold operands (tools/kernel_table.py --only matvec), one box, one run.
# FZ_MATVEC_SLICES: -1 = fp64 accumulation (rounds 1-2: one 1024-thread workgroup per product below 256 columns per CU, a thread
# per column above), 1 / 2 / 4 / 8 / 16 = integer accumulation with that many slices of the k range per column, 0 = the
# library's choice (the one-workgroup-per-product kernel up to two products per CU, the sliced kernel above).
set -e
cd "$(dirname "$0")/../.."
for sp in 256 128; do
  for s in -1 1 2 4 8 16 0; do
    echo "== secpar $sp FZ_MATVEC_SLICES=$s"
    FZ_MATVEC_SLICES=$s python tools/kernel_table.py --only matvec --secpar $sp --matvec-batches 1,16,64,256,512,1024,2048,8192 2>&1 | grep matvec
  done
done
