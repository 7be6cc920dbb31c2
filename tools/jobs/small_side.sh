python -m pytest tests/test_round4_gpu.py tests/test_round5_gpu.py tests/test_engine_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/suite_h1.txt 2>&1; tail -3 gpurun_out/suite_h1.txt | head -2
for wl in c2 with_angle dip; do tools/ab_wl.sh $wl side1 STYLEMESH_SMALL_SIDE=1; tools/ab_wl.sh $wl side0 STYLEMESH_SMALL_SIDE=0; done
tools/ab_wl.sh c3 side1 STYLEMESH_SMALL_SIDE=1
