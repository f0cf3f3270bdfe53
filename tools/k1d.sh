set -e
C=""; for k in 4 8 16 32; do C="$C --case k$k:steps_per_launch=$k"; done
python3 tools/run_kernels.py --shape 1d1r --size 1048576 --launches 50 $C
python3 tools/run_kernels.py --shape 1d1r --size 268435456 --launches 10 $C
python3 tools/run_kernels.py --shape 1d2r --size 1048576 --launches 50 $C
for k in 8 16 32; do python3 bench.py --shape 1d1r --steps 100 --no-cpu-baseline --option steps_per_launch=$k | grep "^{" | cut -c1-120; done
for k in 8 16 32; do python3 bench.py --shape 1d1r --steps 96 --no-cpu-baseline --option steps_per_launch=$k | grep "^{" | cut -c1-120; done
