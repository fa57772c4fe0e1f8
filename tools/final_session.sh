bash tools/gpu_session.sh r05 testsall bench profiles configs timeline stages
O=gpurun_out/r05
python tools/stage_cuts.py 4096 > $O/stage_cuts4096.txt 2>&1
python tools/stage_cuts.py 4096 aliengo perlin > $O/stage_cuts4096_cfg3.txt 2>&1
python tools/stage_cuts.py 4096 go2 flat > $O/stage_cuts4096_cfg4.txt 2>&1
python tools/stage_cuts.py 4096 hyqreal1 random_boxes > $O/stage_cuts4096_cfg5.txt 2>&1
python tools/perf_probe.py stages 4096 aliengo perlin > $O/stages4096_cfg3.txt 2>&1
python tools/perf_probe.py stages 4096 go2 > $O/stages4096_go2.txt 2>&1
python tools/perf_probe.py stages 4096 hyqreal1 random_boxes > $O/stages4096_cfg5.txt 2>&1
tail -3 $O/pytest_gpu.txt
