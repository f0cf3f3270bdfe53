for n in 320 384 448; do
for sh in star3d1r box3d1r; do
python3 tools/run_kernels.py --shape $sh --size $n $n $n --launches 20 --case k1:steps_per_launch=1 --case tile2:stream3=0,steps_per_launch=2 --case k2:steps_per_launch=2 --case k3:steps_per_launch=3 --case k3w4:steps_per_launch=3,stream3_waves=4 --case k3z16:steps_per_launch=3,fused_z_chunk=16 2>/dev/null | grep "^{" | python3 -c "
import sys,json
print('$sh $n', ' '.join(f\"{json.loads(l)['case']}={json.loads(l)['gstencils']}\" for l in sys.stdin))"
done; done
python3 tools/run_kernels.py --shape star3d1r --size 256 512 512 --launches 20 --case tile2:stream3=0,steps_per_launch=2 --case k2:steps_per_launch=2 --case k3:steps_per_launch=3 2>/dev/null | grep "^{" | cut -c1-20,100-170
python3 tools/run_kernels.py --shape star3d1r --size 64 1024 1024 --launches 20 --case tile2:stream3=0,steps_per_launch=2 --case k2:steps_per_launch=2 --case k3:steps_per_launch=3 2>/dev/null | grep "^{" | cut -c1-20,100-170
