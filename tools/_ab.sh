export DT=f32h2 ITERS=20 CHECK=0
for v in f16 bf16 f16 bf16; do
  if [ $v = bf16 ]; then export YV3_LIB=$PWD/yolo_v3_amd/libyv3_expbf16.so; else unset YV3_LIB; fi
  echo "== MFMA $v"; python tools/conv_bench.py c52 c26 L52; echo "== MFMA $v ZERO"; ZERO=1 python tools/conv_bench.py c52 L52; done
