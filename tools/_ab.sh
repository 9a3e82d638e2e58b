export DT=f32h2 ITERS=20 CHECK=0
for k in 0,0,0 0,0,1 0,0,0 0,0,1; do echo "== TUNE=$k (tune[2]=1: no residual touch)"; YV3_TUNE=$k python tools/conv_bench.py c52 c26 c13 c104; done
