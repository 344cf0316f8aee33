#!/bin/bash
# scratch/build_variant_rr.sh <tag> [flags ...]: builds scratch/bin/libsmsut_<tag>.so with conv_wgrad_rr.hip compiled under the
# given extra flags (the other objects come from the product build) -- for in-process A/B runs (scratch/wgrad_rr_ab.py, RR_LIBS).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/smsut-medicalimgsegmentation_amd
mkdir -p $root/scratch/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I $root/include "$@" -c $pkg/csrc/conv_wgrad_rr.hip -o $root/scratch/bin/conv_wgrad_rr_$tag.o 2>/dev/null
objs=$(ls $pkg/lib/*.o | grep -v conv_wgrad_rr.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/scratch/bin/libsmsut_$tag.so $root/scratch/bin/conv_wgrad_rr_$tag.o $objs
echo built $root/scratch/bin/libsmsut_$tag.so
