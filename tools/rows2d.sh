W="--case warm:stream_rows=0"
python3 tools/run_kernels.py --shape star2d1r --size 16384 16384 --launches 10 $W --case new:stream_rows=0 --case r435:stream_rows=431
python3 tools/run_kernels.py --shape star2d1r --size 8192 8192 --launches 20 $W --case new:stream_rows=0 --case r330:stream_rows=330
python3 tools/run_kernels.py --shape box2d3r --size 8192 8192 --launches 20 $W --case new:stream_rows=0 --case r330:stream_rows=330
python3 tools/run_kernels.py --shape star2d3r --size 16384 16384 --launches 10 $W --case new:stream_rows=0 --case r435:stream_rows=431
python3 tools/run_kernels.py --shape star2d1r --size 2144 16384 --launches 20 $W --case new:stream_rows=0 --case r179:stream_rows=179 --case r245:stream_rows=245
python3 tools/run_kernels.py --shape star2d1r --size 4192 16384 --launches 20 $W --case new:stream_rows=0 --case r350:stream_rows=350 --case r233:stream_rows=233
python3 tools/run_kernels.py --shape star2d1r --size 4096 4096 --launches 40 $W --case new:stream_rows=0 --case r400:stream_rows=400 --case r200:stream_rows=200
python3 tools/run_kernels.py --shape star2d1r --size 8192 8191 --launches 20 $W --case new:stream_rows=0 --case r330:stream_rows=330
