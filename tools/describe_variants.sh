#!/bin/bash
# Development aid: builds liborbhip variants that differ in the LDS pitches of k_describe (run here, on the CPU box).
set -e
mkdir -p tools/variants
rm -f tools/variants/index.txt
cd orb_slam2_amd/csrc
OBJS=$(ls *.o | grep -v kernels_describe.o)
i=0
# (source window pitch in bytes, pitch of the row-pair sums in dwords; the kernel checks that the layout they give is aligned and fits)
for v in "-DRAW_P=48 -DHB_P=40" "-DRAW_P=48 -DHB_P=44" "-DRAW_P=52 -DHB_P=40" "-DRAW_P=52 -DHB_P=44" "-DRAW_P=60 -DHB_P=40" "-DRAW_P=60 -DHB_P=44"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $v -c kernels_describe.hip -o ../../tools/variants/desc_$i.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/variants/liborbhip_$i.so $OBJS ../../tools/variants/desc_$i.o -ldl -lpthread
  echo "$i: $v" >> ../../tools/variants/index.txt
  i=$((i+1))
done
