for b in 1024 2048 4608 9216 18432; do python tools/bench_leg.py euroc $b > gpurun_out/leg_euroc_$b.json; python tools/bench_leg.py tum $b > gpurun_out/leg_tum_$b.json; done
