#!/bin/bash
# Development aid: builds liborbhip variants that differ in the LDS pitches of k_describe (run here, on the CPU box).
set -e
cd orb_slam2_amd/csrc
OBJS=$(ls *.o | grep -v kernels_describe.o)
i=0
for v in "-DRAW_P=48 -DHB_P=40 -DBL_P=44" "-DRAW_P=52 -DHB_P=40 -DBL_P=44" "-DRAW_P=48 -DHB_P=44 -DBL_P=44" "-DRAW_P=52 -DHB_P=44 -DBL_P=44" "-DRAW_P=48 -DHB_P=40 -DBL_P=52" "-DRAW_P=52 -DHB_P=44 -DBL_P=52" "-DRAW_P=60 -DHB_P=44 -DBL_P=44" "-DRAW_P=52 -DHB_P=52 -DBL_P=44"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $v -c kernels_describe.hip -o ../../tools/variants/desc_$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/variants/liborbhip_$i.so $OBJS ../../tools/variants/desc_$i.o -ldl -lpthread
  echo "$i: $v" >> ../../tools/variants/index.txt
  i=$((i+1))
done
