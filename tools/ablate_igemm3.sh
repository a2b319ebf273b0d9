#!/bin/bash
# Builds ablated copies of libnqhip.so (conv_igemm3 compiled with -DNQ_IG3_ABL=n; timing experiments, wrong results)
# under tools/_ab/abl<n>/ -- run on the build container; then on the GPU box:
#   for n in 0 1 2 3 4 5 6; do NQ_LIB=tools/_ab/abl$n/libnqhip.so python tools/bench_kernels.py --only igemm3; done
# The weight-gradient kernel has the same kind of switch: compile conv_wgrad3_k{3,5}.hip with -DNQ_WG3_ABL=n
# (conv_wgrad3_impl.h) and link the objects in the same way.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/neuroquant_amd/csrc
make -C $C -j8 > /dev/null
for n in ${@:-1 2 3 4 5 6}; do
  D=$R/tools/_ab/abl$n
  mkdir -p $D
  for k in 3 5; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DNQ_IG3_ABL=$n -c $C/conv_igemm3_k$k.hip -o $D/conv_igemm3_k$k.o &
  done
  wait
  OBJS=$(ls $C/build/*.o | grep -v conv_igemm3_k)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $D/conv_igemm3_k3.o $D/conv_igemm3_k5.o -o $D/libnqhip.so
  echo built $D/libnqhip.so
done
