#!/bin/bash
# scratch/build_alt.sh NAME "-DFLAG ..." : alternate build of conv_mfma.hip linked with the product's other objects
# -> scratch/bin/libsmsut_NAME.so (for the in-process A/B harnesses)
set -e
cd "$(dirname "$0")/.."
P=smsut-medicalimgsegmentation_amd
mkdir -p scratch/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $2 -I include -c $P/csrc/conv_mfma.hip -o scratch/bin/conv_mfma_$1.o
objs=$(ls $P/lib/*.o | grep -v conv_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/bin/libsmsut_$1.so scratch/bin/conv_mfma_$1.o $objs
echo built scratch/bin/libsmsut_$1.so
