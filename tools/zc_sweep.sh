set -e
S=""; for z in 12 16 24 32 48 64; do S="$S --case k3w4z$z:steps_per_launch=3,stream3_waves=4,fused_z_chunk=$z"; done
for z in 16 32 64 0; do S="$S --case k3w8z$z:steps_per_launch=3,stream3_waves=8,fused_z_chunk=$z"; done
B=""; for z in 16 32 64 128 0; do B="$B --case k2w8z$z:steps_per_launch=2,stream3_waves=8,fused_z_chunk=$z --case k2w4z$z:steps_per_launch=2,stream3_waves=4,fused_z_chunk=$z"; done
python3 tools/run_kernels.py --shape star3d1r --size 512 512 512 --launches 10 --case old:stream3=0 $S $B
python3 tools/run_kernels.py --shape star3d1r --size 768 768 768 --launches 6 --case old:stream3=0 $S
python3 tools/run_kernels.py --shape box3d1r --size 768 768 768 --launches 6 --case old:stream3=0 $B --case k3w4z32:steps_per_launch=3,stream3_waves=4,fused_z_chunk=32
python3 tools/run_kernels.py --shape box3d1r --size 512 512 512 --launches 10 --case old:stream3=0 $B
