#!/bin/bash
mkdir -p gpurun_out/r03
O=gpurun_out/r03/two_streams.txt
: > $O
for n in 2 3; do timeout 300 python tools/two_streams.py $n >> $O 2>&1; done
cat $O
