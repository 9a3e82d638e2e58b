#!/bin/bash
# build_variant.sh NAME "-DFLAG ..." : (always with -DYV3_MEASURE: the measurement switches exist only in these builds)
# yolo_v3_amd/libyv3_NAME.so with extra compile flags (kernel A/B on one GPU box via YV3_LIB)
N=$1; F=$2; D=/tmp/var_$N; mkdir -p $D; cd yolo_v3_amd/csrc
for f in conv_igemm_f32 conv_gemm_f32 conv_wino4_f32 conv_planes conv_planes_k3s1 conv_planes_w4 winograd conv0 conv_front conv_front_f32 conv_res64 conv_res64_f32 pack decode postproc prepost gather capi; do
  fl=""; case $f in conv_igemm*|conv_gemm*|conv_wino4*|conv_planes*|conv_front*|conv_res64*|winograd*) ;; *) fl="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function $fl -DYV3_MEASURE $F -c $f.hip -o $D/$f.o 2>/dev/null &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libyv3_$N.so $D/*.o; ls -la ../libyv3_$N.so
