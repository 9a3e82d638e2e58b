export DT=f32h2 ITERS=30 CHECK=0
for t in 0 4 3 0 4; do echo "== YV3_TILE=$t"; YV3_TILE=$t python tools/conv_bench.py p52 p26 p13 c104 p104; done
