set -e
C="--case tile:stream3=0 --case k3z32:steps_per_launch=3 --case k3z64:steps_per_launch=3,fused_z_chunk=64 --case k3z86:steps_per_launch=3,fused_z_chunk=86 --case k3z171:steps_per_launch=3,fused_z_chunk=171 --case k3z256:steps_per_launch=3,fused_z_chunk=256 --case k3w4z32:steps_per_launch=3,stream3_waves=4 --case k3w4z64:steps_per_launch=3,stream3_waves=4,fused_z_chunk=64 --case k3w4z128:steps_per_launch=3,stream3_waves=4,fused_z_chunk=128 --case k2:steps_per_launch=2"
python3 tools/run_kernels.py --shape star3d1r --size 512 512 512 --launches 10 $C
python3 tools/run_kernels.py --shape star3d1r --size 768 768 768 --launches 6 $C
