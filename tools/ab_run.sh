#!/bin/bash
# Same-box A/B of two builds of libfusion_hip.so: boxes of the pool differ by 3-5 % for the same binary, more than most of the
# effects round 5 decided on, so both variants run alternately inside ONE gpurun call and are compared within it.
#   1. build variant A, copy fusion-cryptography_amd/lib/libfusion_hip.so to build/abl/libA.so (build/ travels to the GPU box,
#      is git-ignored); build variant B the same way (or leave it as the in-tree library);
#   2. gpurun -- 'bash tools/ab_run.sh 3 build/abl/libA.so build/abl/libB.so -- python tools/probes/keygen_sign_pair.py'
# Every repetition runs the command once per library (FUSION_HIP_LIB selects it: fusion_hip/_lib.py) and prefixes its output
# lines with the library's name.  "-" as a library = the in-tree one.
set -u
reps=$1; shift
libs=()
while [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
for i in $(seq 1 "$reps"); do
  for lib in "${libs[@]}"; do
    if [ "$lib" = "-" ]; then unset FUSION_HIP_LIB; name=in-tree; else export FUSION_HIP_LIB=$PWD/$lib; name=$(basename "$lib" .so); fi
    "$@" 2>&1 | grep -v "amdgpu.ids" | sed "s/^/$name  /"
  done
done
