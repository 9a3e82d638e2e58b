export TMPDIR=/tmp
O=gpurun_out
for b in 64 48 16; do
  rm -rf $O/pb$b; rocprofv3 --kernel-trace --output-format csv -d $O/pb$b -o t -- python bench.py --no-extras --no-cpu-baseline --steps 10 --batch $b > /dev/null 2> /dev/null
  t=$(find $O/pb$b -name '*kernel_trace.csv' | head -1); python tools/trace_layers.py $t $b 416 > $O/layers_b$b.txt; rm -rf $O/pb$b
done
