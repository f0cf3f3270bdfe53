set -e
C="--case warm:stream3=0 --case tile:stream3=0 --case k2dir:steps_per_launch=2,separable=0 --case k2sep:steps_per_launch=2 --case k3dir:steps_per_launch=3,separable=0 --case k3sep:steps_per_launch=3 --case k3sepw4:steps_per_launch=3,stream3_waves=4 --case k3sepz32:steps_per_launch=3,fused_z_chunk=32"
python3 tools/run_kernels.py --shape box3d1r --size 768 768 768 --launches 6 $C
python3 tools/run_kernels.py --shape box3d1r --size 512 512 512 --launches 10 $C
