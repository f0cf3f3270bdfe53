set -e
C="--case tile:stream3=0 --case k3:steps_per_launch=3 --case k3w4:steps_per_launch=3,stream3_waves=4 --case k2:steps_per_launch=2 --case k3async:steps_per_launch=3,stream3_async=1"
python3 tools/run_kernels.py --shape star3d1r --size 512 512 512 --launches 10 $C
python3 tools/run_kernels.py --shape star3d1r --size 768 768 768 --launches 6 $C
python3 tools/run_kernels.py --shape box3d1r --size 768 768 768 --launches 6 $C
python3 tools/run_kernels.py --shape box3d1r --size 512 512 512 --launches 10 $C
